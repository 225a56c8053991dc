"""GPU parity of the device crop + resize (rn_crop_resize_u8_device, rn_classify_images_u8) with the host
restatement of cv2.resize / RoomNet.center_crop (roomnet_amd/imageops.py, network.py:137-156): byte for byte."""
import numpy as np
import pytest

from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.imageops import resize_linear_u8

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(weights):
    e = _capi.Engine(build_graph(6, 224), weights, device=0, dtype="bf16", max_batch=8)
    yield e
    e.close()


def _center_crop(x):          # network.py:137-146
    h, w, _ = x.shape
    off = abs((w - h) // 2)
    if h < w:
        return x[:, off:off + h, :]
    if w < h:
        return x[off:off + w, :, :]
    return x


def _host(im, side=224):
    c = _center_crop(im)
    return c if c.shape[:2] == (side, side) else resize_linear_u8(c, side, side)


SHAPES = [(224, 224), (448, 448), (448, 600), (601, 448), (480, 640), (1080, 1920), (97, 131), (225, 224), (224, 223),
          (1, 1), (2, 5), (3000, 17), (333, 333), (223, 223), (1024, 768)]


@pytest.mark.parametrize("shape", SHAPES)
def test_crop_resize_is_bit_identical_to_the_host_restatement(engine, shape):
    rng = np.random.default_rng(shape[0] * 10007 + shape[1])
    im = rng.integers(0, 256, (shape[0], shape[1], 3), dtype=np.uint8)
    got = engine.crop_resize(im)
    want = _host(im)
    assert got.shape == (224, 224, 3)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("shape", [(37, 53), (61, 40), (300, 301)])
def test_crop_resize_vs_the_scalar_oracle(engine, shape):
    """Against the independent loop-per-pixel restatement under oracle/ (not the NumPy product code)."""
    from oracle.cv_resize_ref import center_crop, resize_linear_u8_scalar
    rng = np.random.default_rng(shape[0] + 1000 * shape[1])
    im = rng.integers(0, 256, (shape[0], shape[1], 3), dtype=np.uint8)
    want = resize_linear_u8_scalar(center_crop(im), 224, 224)
    np.testing.assert_array_equal(engine.crop_resize(im), want)


def test_structured_images_and_extremes(engine):
    yy, xx = np.mgrid[0:517, 0:389]
    grad = np.stack([(xx * 255 // 388), (yy * 255 // 516), ((xx + yy) % 256)], -1).astype(np.uint8)
    for im in (grad, np.zeros((300, 500, 3), np.uint8), np.full((500, 300, 3), 255, np.uint8),
               (np.indices((640, 480)).sum(0) % 2 * 255).astype(np.uint8)[:, :, None].repeat(3, 2)):
        np.testing.assert_array_equal(engine.crop_resize(im), _host(im))


def test_classify_images_matches_host_pipeline_then_forward(engine):
    rng = np.random.default_rng(7)
    ims = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for (h, w) in [(300, 400), (224, 224), (448, 448), (500, 250),
                                                                           (231, 999), (64, 64), (720, 1280), (225, 225),
                                                                           (333, 222), (100, 101)]]      # > max_batch: chunks
    ids, probs = engine.classify_images(ims)
    host_batch = np.stack([_host(im) for im in ims])
    ids2, probs2 = engine.forward_u8(host_batch)
    np.testing.assert_array_equal(ids, ids2)
    np.testing.assert_array_equal(probs, probs2)


def test_bad_images_are_rejected(engine):
    with pytest.raises(ValueError):
        engine.classify_images([np.zeros((10, 10), np.uint8)])
    with pytest.raises(ValueError):
        engine.classify_images([np.zeros((0, 10, 3), np.uint8)])
    buf = np.zeros((4, 4, 3), np.uint8)
    d = engine.device_malloc(64)
    try:
        rc = engine.lib.rn_crop_resize_u8_device(engine.handle, d, 4, 4, d, engine.max_batch)    # slot out of range
        assert rc < 0 and b"rn_crop_resize_u8_device" in engine.lib.rn_last_error()
    finally:
        engine.device_free(d)


def test_batched_crop_resize_is_one_launch_and_byte_identical(engine):
    """rn_crop_resize_batch_u8_device: eight images of eight different shapes (copy, exact 2x box, up- and down-scales, 1 x 1) in ONE
    launch, every slot byte for byte the host restatement -- and twice in a row on the same handle (the table is re-uploaded)."""
    shapes = [(224, 224), (448, 448), (480, 640), (1080, 1920), (97, 131), (1, 1), (3000, 17), (601, 448)]
    rng = np.random.default_rng(77)
    ims = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    for order in (range(8), reversed(range(8))):
        batch = [ims[k] for k in order]
        got = engine.crop_resize_batch(batch)
        assert got.shape == (8, 224, 224, 3)
        for k, im in enumerate(batch):
            np.testing.assert_array_equal(got[k], _host(im), err_msg=str(im.shape))
    with pytest.raises((ValueError, _capi.RoomNetLibraryError)):
        engine.crop_resize_batch([ims[0]] * 9)           # more than max_batch
