"""GPU box: images/sec of the caller-side pipeline.

  python tools/bench_images.py [H W [N]]          N decoded images of one size (decode excluded):
        host centre-crop + resize (roomnet_amd.imageops) + rn_forward_u8   vs   rn_classify_images_u8 (crop + resize on the GPU)
  python tools/bench_images.py --dir [--threads=T] [H W [N]]    a generated directory of N JPEG files of that size through
        classify_im_dir(overlay=False) and groundtruth_validation (decode on the thread pool, crop + resize + forward on
        the GPU), next to decode alone and to the one-image-at-a-time loop of the reference's caller (infer.py:79-82)
"""
import contextlib
import io
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, '.')
from roomnet_amd import _capi, imageio
from roomnet_amd.graph import build_graph
from roomnet_amd.imageops import resize_linear_u8
from roomnet_amd.tf_bundle import BundleReader

THREADS = None
for _a in sys.argv[1:]:
    if _a.startswith('--threads='):
        THREADS = int(_a.split('=', 1)[1])          # decode threads of the directory drivers (default: infer.DECODE_THREADS)
args = [a for a in sys.argv[1:] if not a.startswith('--')]
H = int(args[0]) if len(args) > 0 else 1080
W = int(args[1]) if len(args) > 1 else 1920
N = int(args[2]) if len(args) > 2 else 64


def crop(x):
    h, w_, _ = x.shape
    off = abs((w_ - h) // 2)
    return x[:, off:off + h, :] if h < w_ else (x[off:off + w_, :, :] if w_ < h else x)


def decoded_images():
    w = BundleReader('roomnet_amd/final_model/roomnet').load_all()
    e = _capi.Engine(build_graph(6, 224), w, dtype='bf16', max_batch=N)
    rng = np.random.default_rng(0)
    ims = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(N)]
    e.classify_images(ims[:2])
    t0 = time.perf_counter()
    ids_g, p_g = e.classify_images(ims)
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    batch = np.stack([resize_linear_u8(crop(im), 224, 224) for im in ims])
    ids_h, p_h = e.forward_u8(batch)
    t_host = time.perf_counter() - t0
    assert (ids_g == ids_h).all() and (p_g == p_h).all()
    print('%d images %dx%d: GPU crop+resize+forward %.1f img/s (%.2f ms/img, %.0f MB of crop windows uploaded)   '
          'host crop+resize + forward %.1f img/s (%.1f ms/img)' % (
              N, W, H, N / t_gpu, 1e3 * t_gpu / N, N * min(H, W) ** 2 * 3 / 1e6, N / t_host, 1e3 * t_host / N))


def directory():
    from roomnet_amd import infer
    from roomnet_amd.network import RoomNet
    if THREADS:
        infer.DECODE_THREADS = THREADS
    root = tempfile.mkdtemp(prefix='rn_bench_')
    d = os.path.join(root, 'images')
    os.makedirs(d)
    rng = np.random.default_rng(0)
    # photograph-like content (smooth fields + a little noise) so the JPEG files have a realistic size
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    for k in range(N):
        f = rng.uniform(0.002, 0.02, 6)
        im = np.stack([127 + 100 * np.sin(f[2 * c] * xx + k) * np.cos(f[2 * c + 1] * yy) for c in range(3)], -1)
        im = np.clip(im + rng.normal(0, 6, im.shape), 0, 255).astype(np.uint8)
        imageio.imwrite(os.path.join(d, 'im_%04d.jpg' % k), im)
    mb = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) / 1e6
    nn = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, optimized_inference=True, dtype='bf16', max_batch=64)
    nn.load(os.path.join('roomnet_amd', 'final_model', 'roomnet'))
    paths = sorted(os.path.join(d, f) for f in os.listdir(d))
    nn.infer_images([imageio.imread(paths[0])])
    t0 = time.perf_counter()
    for p in paths:
        imageio.imread(p)
    t_dec1 = time.perf_counter() - t0
    import sklearn.metrics  # noqa: F401  (groundtruth_validation imports it inside the call: keep the one-off import out of its time)
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        t0 = time.perf_counter()
        infer.classify_im_dir(nn, d, overlay=False, batch_size=64)
        t_dir = time.perf_counter() - t0
        t0 = time.perf_counter()
        infer.classify_im_dir(nn, d, overlay=True, batch_size=64)          # the reference's default: overlay + re-encoded files
        t_ovl = time.perf_counter() - t0
        lst = os.path.join(root, 'list.txt')
        with open(lst, 'w') as f:
            f.writelines('%s %d\n' % (p, 0) for p in paths)
        t0 = time.perf_counter()
        infer.groundtruth_validation(nn, lst, batch_size=64)
        t_val = time.perf_counter() - t0
        # the reference's caller: one image per call (infer.py:79-82), host decode, serial
        t0 = time.perf_counter()
        for p in paths[:max(8, N // 8)]:
            nn.infer_optimized(imageio.imread(p))
        t_one = (time.perf_counter() - t0) / max(8, N // 8)
    # what the overlay path cost per image while it ran inside the loop (two put_text + imwrite, one thread)
    ims = [imageio.imread(p) for p in paths[:max(8, N // 16)]]
    t0 = time.perf_counter()
    for k, im in enumerate(ims):
        infer._overlay_and_write(im, 'LivingRoom', np.float32(0.9987), os.path.join(root, 'ovl_%d.jpg' % k))
    t_ow = (time.perf_counter() - t0) / len(ims)
    print('%d JPEG files %dx%d (%.0f MB), %d decode threads of %d host cores: classify_im_dir(overlay=False) %.1f img/s   '
          'classify_im_dir(overlay=True) %.1f img/s (overlay + write alone, one thread: %.1f img/s)   '
          'groundtruth_validation %.1f img/s   decode alone, one thread %.1f img/s   one image per call (reference loop) %.1f img/s'
          % (N, W, H, mb, infer.DECODE_THREADS, os.cpu_count() or 1, N / t_dir, N / t_ovl, 1.0 / t_ow, N / t_val, N / t_dec1, 1.0 / t_one))
    shutil.rmtree(root, ignore_errors=True)


if '--dir' in sys.argv:
    directory()
else:
    decoded_images()
