"""Host-only timing of the directory drivers' decode step (roomnet_amd.imageio.imread) on this machine: the parts of one decode and the
thread scaling of the pool.  usage: python tools/bench_decode.py [H W N]"""
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from roomnet_amd import imageio  # noqa: E402

H, W, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (1080, 1920, 64)))


def main():
    from PIL import Image
    d = tempfile.mkdtemp(prefix='rn_dec_')
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    paths = []
    for k in range(N):
        f = rng.uniform(0.002, 0.02, 6)
        im = np.stack([127 + 100 * np.sin(f[2 * c] * xx + k) * np.cos(f[2 * c + 1] * yy) for c in range(3)], -1)
        im = np.clip(im + rng.normal(0, 6, im.shape), 0, 255).astype(np.uint8)
        p = os.path.join(d, 'im_%04d.jpg' % k)
        imageio.imwrite(p, im)
        paths.append(p)
    t = np.zeros(5)
    for p in paths:
        t0 = time.perf_counter()
        im = Image.open(p)
        im.load()
        t1 = time.perf_counter()
        a = np.asarray(im, dtype=np.uint8)
        t2 = time.perf_counter()
        b = np.ascontiguousarray(a[:, :, ::-1])
        t3 = time.perf_counter()
        c = np.ascontiguousarray(a[:, :, [2, 1, 0]])
        t4 = time.perf_counter()
        e = np.frombuffer(bytearray(im.tobytes('raw', 'BGR')), np.uint8).reshape(a.shape)
        t5 = time.perf_counter()
        assert np.array_equal(b, c) and np.array_equal(b, e)
        t += (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)
    print('%d x %d JPEG, ms per image: decode %.2f  asarray %.2f  swap by negative stride %.2f  by index list %.2f  '
          'Pillow raw BGR (instead of asarray + swap) %.2f' % ((W, H) + tuple(t / N * 1e3)))
    for n in (1, 2, 4, 8, 16, 32, 64):
        if n > 2 * (os.cpu_count() or 1):
            break
        with ThreadPoolExecutor(n) as ex:
            t0 = time.perf_counter()
            list(ex.map(imageio.imread, paths * 4))
            dt = time.perf_counter() - t0
        print('  %2d threads: %.1f img/s' % (n, 4 * N / dt))


if __name__ == '__main__':
    main()
