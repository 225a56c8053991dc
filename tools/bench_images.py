"""GPU box: end-to-end images/sec of the caller pipeline for N images of one size (decode excluded):
host centre-crop + resize (roomnet_amd.imageops) + rn_forward_u8   vs   rn_classify_images_u8 (crop + resize on the GPU).
usage: python tools/bench_images.py [H W [N]]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader
from roomnet_amd.imageops import resize_linear_u8
H = int(sys.argv[1]) if len(sys.argv) > 1 else 1080
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
N = int(sys.argv[3]) if len(sys.argv) > 3 else 64
w = BundleReader('roomnet_amd/final_model/roomnet').load_all()
e = _capi.Engine(build_graph(6, 224), w, dtype='bf16', max_batch=N)
rng = np.random.default_rng(0)
ims = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(N)]
def crop(x):
    h, w_, _ = x.shape; off = abs((w_ - h) // 2)
    return x[:, off:off + h, :] if h < w_ else (x[off:off + w_, :, :] if w_ < h else x)
e.classify_images(ims[:2])
t0 = time.perf_counter(); ids_g, p_g = e.classify_images(ims); t_gpu = time.perf_counter() - t0
t0 = time.perf_counter()
batch = np.stack([resize_linear_u8(crop(im), 224, 224) for im in ims]); ids_h, p_h = e.forward_u8(batch)
t_host = time.perf_counter() - t0
assert (ids_g == ids_h).all() and (p_g == p_h).all()
print('%d images %dx%d: GPU crop+resize+forward %.1f img/s (%.2f ms/img, %.0f MB of crop windows uploaded)   host crop+resize + forward %.1f img/s (%.1f ms/img)' % (
    N, W, H, N / t_gpu, 1e3 * t_gpu / N, N * min(H, W) ** 2 * 3 / 1e6, N / t_host, 1e3 * t_host / N))
