"""CPU tests of the host-side pieces around the hot path: OpenCV-style resize restatement,
xls writer, image I/O, list-file parsing, checkpoint save/load semantics of RoomNet."""
import os

import numpy as np
import pytest

from roomnet_amd import imageio, imageops, xls
from xls_reader import read_xls


# ------------------------------------------------------------------ resize (cv2.resize INTER_LINEAR, uint8)
from oracle.cv_resize_ref import resize_linear_u8_scalar as _brute_force_linear  # noqa: E402


@pytest.mark.parametrize("shape,dst", [((37, 53, 3), (24, 24)), ((20, 20, 3), (33, 33)), ((50, 31, 1), (17, 40))])
def test_resize_matches_scalar_restatement(shape, dst):
    rng = np.random.default_rng(sum(shape))
    src = rng.integers(0, 256, shape, dtype=np.uint8)
    got = imageops.resize_linear_u8(src, dst[0], dst[1])
    np.testing.assert_array_equal(got, _brute_force_linear(src, dst[0], dst[1]))


def test_resize_properties():
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (48, 48, 3), dtype=np.uint8)
    same = imageops.resize_linear_u8(src, 48, 48)
    np.testing.assert_array_equal(same, src)
    assert same is not src
    flat = np.full((31, 57, 3), 77, np.uint8)
    assert (imageops.resize_linear_u8(flat, 224, 224) == 77).all()        # constants are preserved
    half = imageops.resize_linear_u8(src, 24, 24)                          # exact 2x: 2x2 box average
    s = src.astype(np.int32)
    box = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    np.testing.assert_array_equal(half, box.astype(np.uint8))
    gray = imageops.resize_linear_u8(src[:, :, 0], 10, 12)
    assert gray.shape == (12, 10)
    with pytest.raises(TypeError):
        imageops.resize_linear_u8(src.astype(np.float32), 8, 8)
    up = imageops.resize_linear_u8(src, 96, 96)
    assert up.min() >= src.min() and up.max() <= src.max()                  # convex combination


# ------------------------------------------------------------------ xls writer
def test_xls_round_trip(tmp_path):
    wb = xls.Workbook()
    sh = wb.add_sheet("classification_results")
    sh.write(0, 0, "IMAGE_NAME")
    sh.write(0, 1, "PREDICTED_LABEL")
    rows = [("a b.jpg", "Kitchen", "0.9312"), ("ünï.png", "Bedroom", "0.5"), ("c.png", "Kitchen", "1.0")]
    for i, (a, b, c) in enumerate(rows):
        sh.write(i + 1, 0, a)
        sh.write(i + 1, 1, b)
        sh.write(i + 1, 2, c)
    sh.write(5, 0, 12.5)
    p = str(tmp_path / "out.xls")
    wb.save(p)
    got = read_xls(p)
    assert list(got) == ["classification_results"]
    cells = got["classification_results"]
    assert cells[(0, 0)] == "IMAGE_NAME" and cells[(0, 1)] == "PREDICTED_LABEL"
    for i, row in enumerate(rows):
        assert tuple(cells[(i + 1, k)] for k in range(3)) == row
    assert cells[(5, 0)] == 12.5
    assert os.path.getsize(p) % 512 == 0
    with pytest.raises(Exception):
        sh.write(0, 0, "again")                                             # xlwt refuses overwrites too


def test_xls_many_rows_and_sheets(tmp_path):
    wb = xls.Workbook()
    s1 = wb.add_sheet("one")
    s2 = wb.add_sheet("two")
    for i in range(3000):
        s1.write(i, 0, "file_%05d.jpg" % i)
        s1.write(i, 1, "label%d" % (i % 6))
    s2.write(0, 0, "x")
    p = str(tmp_path / "big.xls")
    wb.save(p)
    got = read_xls(p)
    assert got["one"][(2999, 0)] == "file_02999.jpg" and got["one"][(1234, 1)] == "label4"
    assert got["two"] == {(0, 0): "x"}
    with pytest.raises(Exception):
        wb.add_sheet("ONE")
    with pytest.raises(ValueError):
        wb.add_sheet("bad/name")


# ------------------------------------------------------------------ image I/O
def test_imread_imwrite_bgr_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    im = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    p = str(tmp_path / "x.png")
    assert imageio.imwrite(p, im)
    back = imageio.imread(p)
    np.testing.assert_array_equal(back, im)                                 # PNG is lossless, BGR order kept
    from PIL import Image
    assert tuple(Image.open(p).getpixel((5, 7))) == tuple(int(v) for v in im[7, 5, ::-1])
    Image.fromarray(im[:, :, 0]).save(str(tmp_path / "g.png"))               # grayscale -> 3 equal channels
    g = imageio.imread(str(tmp_path / "g.png"))
    assert g.shape == (40, 60, 3) and (g[..., 0] == g[..., 2]).all()
    rgba = np.dstack([im[:, :, ::-1], np.full((40, 60), 128, np.uint8)])
    Image.fromarray(rgba, "RGBA").save(str(tmp_path / "a.png"))
    np.testing.assert_array_equal(imageio.imread(str(tmp_path / "a.png")), im)   # alpha dropped
    (tmp_path / "junk.jpg").write_bytes(b"not an image")
    assert imageio.imread(str(tmp_path / "junk.jpg")) is None
    assert imageio.imread(str(tmp_path / "missing.png")) is None
    jp = str(tmp_path / "x.jpg")
    assert imageio.imwrite(jp, im) and imageio.imread(jp).shape == im.shape


def test_imread_zero_copy_pixel_export_equals_the_plain_path(tmp_path, monkeypatch):
    """imread takes the pixels through Pillow's Arrow export when it can (no GIL-held copy) and through np.asarray otherwise:
    the same array either way, writable and contiguous (the overlay draws into it), also for an odd width."""
    from PIL import Image
    rng = np.random.default_rng(11)
    for k, (h, w) in enumerate([(37, 53), (480, 640), (1, 1)]):
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = str(tmp_path / ("z%d.png" % k))
        assert imageio.imwrite(p, im)
        with Image.open(p) as pim:
            pim.load()
            view = imageio._pixels_zero_copy(pim)
            if hasattr(pim, "__arrow_c_array__") and imageio._pa:
                assert view is not None and view.shape == (h, w, 4)
                np.testing.assert_array_equal(view[:, :, :3], np.asarray(pim))
        fast = imageio.imread(p)
        monkeypatch.setattr(imageio, "_pixels_zero_copy", lambda im_: None)
        plain = imageio.imread(p)
        monkeypatch.undo()
        np.testing.assert_array_equal(fast, im)
        np.testing.assert_array_equal(plain, im)
        assert fast.flags.c_contiguous and fast.flags.writeable and fast.dtype == np.uint8


def test_put_text_draws_in_the_requested_colour():
    im = np.zeros((720, 1280, 3), np.uint8)
    imageio.put_text(im, "Predicted Class: Kitchen", (640, 648), (720 / 720.) * .85, (0, 255, 0))
    assert im[..., 1].max() == 255 and im[..., 0].max() == 0 and im[..., 2].max() == 0
    ys, xs = np.nonzero(im[..., 1])
    assert xs.min() >= 640 and ys.max() <= 660 and ys.min() >= 600          # anchored at org (bottom-left)


# ------------------------------------------------------------------ list files
def test_read_fpaths(tmp_path):
    pytest.importorskip("ctypes")
    from roomnet_amd import _capi
    if not os.path.isfile(_capi.LIB_PATH):
        pytest.skip("library not built")
    from roomnet_amd.infer import read_fpaths, CLASS_LABELS, IMG_SIDE
    p = tmp_path / "list.txt"
    p.write_text("C:\\data\\living room\\a b.jpg 5\nrel/x.png 0\n")
    paths, ids, n = read_fpaths(str(p))
    assert paths == ["C:\\data\\living room\\a b.jpg", "rel/x.png"] and ids == [5, 0] and n == 2
    assert CLASS_LABELS == ['Backyard', 'Bathroom', 'Bedroom', 'Frontyard', 'Kitchen', 'LivingRoom']
    assert IMG_SIDE == 224


# ------------------------------------------------------------------ RoomNet persistence semantics (no GPU needed)
def test_roomnet_init_load_save_semantics(tmp_path, weights, capsys, monkeypatch):
    from roomnet_amd import _capi
    if not os.path.isfile(_capi.LIB_PATH):
        pytest.skip("library not built")
    from roomnet_amd.network import RoomNet
    from roomnet_amd import tf_bundle
    from conftest import MODEL_PREFIX
    nn = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, optimized_inference=True)
    assert nn.sess is None and nn.num_classes == 6 and nn.im_side == 224
    nn.load(MODEL_PREFIX)
    assert "Model restored from" in capsys.readouterr().out
    for k, v in weights.items():
        np.testing.assert_array_equal(nn.sess.variables[k], v)
    # save() in optimized mode writes ./roomnet.{index,data-...} like the reference (network.py:94-97)
    monkeypatch.chdir(tmp_path)
    nn.save()
    assert "Model Saved in optimized inference mode" in capsys.readouterr().out
    r = tf_bundle.BundleReader(str(tmp_path / "roomnet"))
    assert sorted(r.keys()) == sorted(weights)
    # load(None): newest step wins (network.py:108-118)
    os.makedirs("all_trained_models/trained_models")
    for step, scale in ((100, 1.0), (2500, 2.0), (900, 3.0)):
        w2 = {k: (v * scale).astype(np.float32) for k, v in weights.items()}
        tf_bundle.write_bundle("all_trained_models/trained_models/roomnet--0.5--%d" % step, w2)
    nn2 = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, optimized_inference=True)
    nn2.load()
    assert nn2.step == 2500
    np.testing.assert_array_equal(nn2.sess.variables["conv2d/kernel"], weights["conv2d/kernel"] * 2.0)
    # training-mode restorer skips the dense blocks (restore_excluded_vars, network.py:242)
    nn3 = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False)
    nn3.load(MODEL_PREFIX)
    np.testing.assert_array_equal(nn3.sess.variables["conv2d_9/kernel"], weights["conv2d_9/kernel"])
    assert not np.array_equal(nn3.sess.variables["dense/kernel"], weights["dense/kernel"])
    # nothing to restore from
    monkeypatch.chdir(tmp_path / "all_trained_models")
    nn4 = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, optimized_inference=True)
    nn4.load()
    assert "No model found to restore from, initializing random weights" in capsys.readouterr().out
    # wrong im_side for the checkpoint: dense/kernel shape mismatch, like TF's restore error
    nn5 = RoomNet(num_classes=6, im_side=600, compute_bn_mean_var=False, optimized_inference=True)
    with pytest.raises(ValueError):
        nn5.load(MODEL_PREFIX)
    with pytest.raises(IOError):
        nn.load(str(tmp_path / "does_not_exist"))
    with pytest.raises(NotImplementedError):
        RoomNet(num_classes=6, im_side=224)            # compute_bn_mean_var=True is the training path
    with pytest.raises(NotImplementedError):
        nn.train_step(None, None)
    x = np.arange(4 * 7 * 3).reshape(4, 7, 3)
    np.testing.assert_array_equal(nn.center_crop(x), x[:, 1:5, :])


def test_imread_16bit_grey_takes_the_high_byte(tmp_path):
    """cv2.imread(path) (IMREAD_COLOR) strips 16-bit samples to their high byte; Pillow's convert() would clip."""
    from PIL import Image
    from roomnet_amd.imageio import imread
    a = (np.arange(64 * 48, dtype=np.uint32).reshape(48, 64) * 21 % 65536).astype(np.uint16)
    p = str(tmp_path / "g16.png")
    Image.fromarray(a).save(p)
    im = imread(p)
    assert im.shape == (48, 64, 3) and im.dtype == np.uint8
    np.testing.assert_array_equal(im[:, :, 0], (a >> 8).astype(np.uint8))
    np.testing.assert_array_equal(im[:, :, 1], im[:, :, 2])


def test_xls_workbook_structure_follows_ms_xls(tmp_path):
    """Structural check of the BIFF8 stream against [MS-XLS] -- independent of tests/xls_reader.py's cell decoding:
    globals substream BOF(0x0005) ... EOF, one BOUNDSHEET8 per sheet whose lbPlyPos points at that sheet's BOF(0x0010),
    SST total / unique counts, DIMENSIONS bounds, every LABELSST index inside the SST, BOF/EOF strictly paired."""
    import struct
    from roomnet_amd import xls
    from xls_reader import workbook_stream
    wb = xls.Workbook()
    sh = wb.add_sheet("classification_results")
    sh.write(0, 0, "IMAGE_NAME")
    sh.write(0, 1, "PREDICTED_LABEL")
    names = ["img_%03d.png" % i for i in range(300)]            # > 8224 bytes of strings: SST spills into CONTINUE
    for i, nm in enumerate(names):
        sh.write(i + 1, 0, nm)
        sh.write(i + 1, 1, ["Kitchen", "Bedroom"][i % 2])
        sh.write(i + 1, 2, str(np.float32(0.5 + i / 1000.0)))
    p = str(tmp_path / "s.xls")
    wb.save(p)
    stream = workbook_stream(p)
    recs, off = [], 0
    while off < len(stream):
        rid, ln = struct.unpack_from("<HH", stream, off)
        recs.append((rid, off, stream[off + 4:off + 4 + ln]))
        off += 4 + ln
        assert ln <= 8224, "record %04x longer than the BIFF8 limit" % rid
    assert off == len(stream)
    ids = [r[0] for r in recs]
    # BOF / EOF pairing: globals substream first, then one worksheet substream
    assert ids[0] == 0x0809 and struct.unpack_from("<HH", recs[0][2])[:2] == (0x0600, 0x0005)
    depth, substreams = 0, []
    for rid, roff, body in recs:
        if rid == 0x0809:
            assert depth == 0
            depth = 1
            substreams.append((struct.unpack_from("<H", body, 2)[0], roff))
        elif rid == 0x000A:
            assert depth == 1
            depth = 0
    assert depth == 0 and [t for t, _ in substreams] == [0x0005, 0x0010]
    # BOUNDSHEET8: stream offset of the sheet's BOF, visible worksheet, name
    bs = [body for rid, _, body in recs if rid == 0x0085]
    assert len(bs) == 1
    lbplypos, hs, dt, cch, fhigh = struct.unpack_from("<IBBBB", bs[0])
    assert lbplypos == substreams[1][1] and hs == 0 and dt == 0
    nm = bs[0][8:8 + cch * (2 if fhigh else 1)].decode("utf-16-le" if fhigh else "latin-1")
    assert nm == "classification_results"
    # SST: total references and unique strings
    sst = [body for rid, _, body in recs if rid == 0x00FC]
    assert len(sst) == 1
    total, unique = struct.unpack_from("<II", sst[0])
    n_label = sum(1 for rid in ids if rid == 0x00FD)
    assert total == n_label == 2 + 3 * 300
    assert unique == 2 + 300 + 2 + 300                          # headers + names + 2 labels + 300 distinct confidences
    assert any(rid == 0x003C for rid in ids), "an SST this large must continue in CONTINUE records"
    for rid, _, body in recs:
        if rid == 0x00FD:
            r, c, xf, isst = struct.unpack_from("<HHHI", body)
            assert isst < unique and r <= 300 and c <= 2
    # DIMENSIONS of the sheet: rows [0, 301), columns [0, 3)
    dim = [body for rid, roff, body in recs if rid == 0x0200 and roff > substreams[1][1]]
    assert struct.unpack_from("<IIHH", dim[0]) == (0, 301, 0, 3)


def test_overlay_text_is_hershey_simplex_strokes_at_the_cv2_anchor():
    """cv2.putText(img, text, org, FONT_HERSHEY_SIMPLEX, scale, color, 1, LINE_AA) (reference infer.py:89-92): org is the
    bottom-left corner on the baseline, font unit (x, y) lands on (org.x + x * scale, org.y - y * scale), capitals are
    21 units tall, descenders reach 7 units below, a glyph advances the pen by its table width."""
    from roomnet_amd import hershey
    from roomnet_amd.imageio import put_text
    scale = 2.0
    im = np.zeros((160, 700, 3), np.uint8)
    org = (20, 100)
    put_text(im, "Predicted Class: Bedroom", org, scale, (0, 255, 0))
    ys, xs = np.nonzero(im[:, :, 1])
    assert (im[:, :, 0] == 0).all() and (im[:, :, 2] == 0).all()            # pure green
    assert abs(ys.min() - (org[1] - 22 * scale)) <= 1.5                       # capitals reach 21 units, the dot of 'i' 22
    assert ys.max() <= org[1] + 1                                             # nothing in this text descends
    assert abs(xs.min() - (org[0] + 4 * scale)) <= 1.5                        # 'P' starts at font x = 4
    assert xs.max() <= org[0] + hershey.text_width("Predicted Class: Bedroom", scale)
    assert ((im[:, :, 1] > 0) & (im[:, :, 1] < 255)).any()                    # anti-aliased edges
    im2 = np.zeros((160, 300, 3), np.uint8)
    put_text(im2, "gy", (10, 100), scale, (255, 255, 255))
    assert abs(np.nonzero(im2[:, :, 0])[0].max() - (100 + 7 * scale)) <= 1.5  # descenders: 7 units below the baseline
    # a one-stroke glyph: 'l' is the segment (4,21)-(4,0): a vertical line one pixel wide at x = org.x + 4 * scale
    im3 = np.zeros((80, 40, 3), np.uint8)
    put_text(im3, "l", (5, 60), 1.0, (0, 0, 255))
    col = im3[:, :, 2].sum(0)
    assert col.argmax() in (9, 10) and (col > 0).sum() <= 3
    rows = np.nonzero(im3[:, :, 2].sum(1))[0]
    assert abs(rows.min() - 39) <= 1.5 and abs(rows.max() - 60) <= 1.5
    # characters outside the table are drawn as '?', like OpenCV does; text off the image is clipped, not an error
    a, b = np.zeros((60, 60, 3), np.uint8), np.zeros((60, 60, 3), np.uint8)
    put_text(a, "é", (5, 50), 1.0, (9, 9, 9))
    put_text(b, "?", (5, 50), 1.0, (9, 9, 9))
    np.testing.assert_array_equal(a, b)
    put_text(a, "Kitchen", (50, 500), 1.0, (1, 2, 3))
    assert len(hershey.GLYPHS) >= 76 and all(len(s) >= 2 for _, strokes in hershey.GLYPHS.values() for s in strokes)


def test_infer_files_decodes_on_a_pool_and_keeps_list_order(tmp_path, capsys):
    """The driver loop without a GPU: files are decoded on the thread pool, unreadable ones are reported and skipped,
    results come back in list order whatever the batch size, and a model object without ``infer_images`` (the
    reference's own class) gets host-prepared batches through ``infer``."""
    from roomnet_amd import imageio
    from roomnet_amd.infer import _infer_files
    from roomnet_amd.imageops import resize_linear_u8

    class Stub:
        im_side = 8
        batches = []

        def center_crop(self, x):
            h, w, _ = x.shape
            o = abs((w - h) // 2)
            return x[:, o:o + h, :] if h < w else (x[o:o + w, :, :] if w < h else x.copy())

        def infer(self, batch):
            assert batch.dtype == np.uint8 and batch.shape[1:] == (8, 8, 3)
            Stub.batches.append(len(batch))
            ids = batch[:, 0, 0, 0].astype(np.int64) % 6                  # "class" = a pixel of the prepared image
            probs = np.zeros((len(batch), 6), np.float32)
            probs[np.arange(len(batch)), ids] = 0.5
            return ids, probs

    rng = np.random.default_rng(3)
    paths, want = [], []
    for k in range(11):
        p = tmp_path / ("f_%02d.png" % k)
        if k in (4, 9):
            p.write_bytes(b"junk")
        else:
            h, w = int(rng.integers(8, 40)), int(rng.integers(8, 40))
            im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            assert imageio.imwrite(str(p), im)
            c = Stub().center_crop(im)
            prep = c if c.shape[0] == 8 else resize_linear_u8(np.ascontiguousarray(c), 8, 8)
            want.append((k, int(prep[0, 0, 0]) % 6))
        paths.append(str(p))
    for bs, threads in ((3, 4), (64, 1)):
        Stub.batches = []
        got = [(i, idx) for i, _im, idx, _conf in _infer_files(Stub(), paths, bs, decode_threads=threads)]
        assert got == want
        assert Stub.batches == ([3, 3, 3] if bs == 3 else [9])
    assert capsys.readouterr().out.count("unreadable image, skipped") == 4


def test_classify_im_dir_writes_overlays_on_the_writer_pool(tmp_path, capsys):
    """infer.py:65-100 without a GPU (a model object with the reference's ``infer`` only): every readable image lands in its
    class directory with the two overlay lines drawn into it, the workbook rows follow the loop's order, and the function
    returns after the writer pool has written the last file."""
    import glob as _glob
    from roomnet_amd import imageio
    from roomnet_amd.infer import CLASS_LABELS, classify_im_dir

    class Stub:
        im_side = 8

        def center_crop(self, x):
            h, w, _ = x.shape
            o = abs((w - h) // 2)
            return x[:, o:o + h, :] if h < w else (x[o:o + w, :, :] if w < h else x.copy())

        def infer(self, batch):
            ids = batch[:, 0, 0, 0].astype(np.int64) % 6
            probs = np.full((len(batch), 6), 0.1, np.float32)
            probs[np.arange(len(batch)), ids] = 0.5
            return ids, probs

    d = tmp_path / "imgs"
    d.mkdir()
    rng = np.random.default_rng(8)
    for k in range(23):
        im = np.full((120, 200, 3), int(rng.integers(20, 60)), np.uint8)       # flat and dark: the overlay is what stands out
        im[:8, :, :] = rng.integers(0, 256, (8, 200, 3), dtype=np.uint8)
        assert imageio.imwrite(str(d / ("p_%02d.png" % k)), im)
    (d / "junk.png").write_bytes(b"no image")
    xl = classify_im_dir(Stub(), str(d), overlay=True, batch_size=4)
    out = capsys.readouterr().out
    listed = [ln.split(" ---> ")[0] for ln in out.splitlines() if " ---> " in ln and "unreadable" not in ln]
    assert len(listed) == 23 and "unreadable" in out
    written = sorted(_glob.glob(str(tmp_path / "imgs_classified" / "*" / "*.png")))
    assert sorted(os.path.basename(p) for p in written) == sorted(os.path.basename(p) for p in listed)
    for p in written:
        label = os.path.basename(os.path.dirname(p))
        assert label in CLASS_LABELS
        got, src = imageio.imread(p), imageio.imread(str(d / os.path.basename(p)))
        assert got.shape == src.shape
        changed = np.abs(got.astype(int) - src.astype(int)).max(axis=2) > 40
        assert changed[90:, 90:].sum() > 30 and not changed[:60].any()             # text in the lower right half only
        green, blue = got[..., 1].astype(int) - got[..., 0], got[..., 0].astype(int) - got[..., 1]
        assert (green[changed] > 100).any() and (blue[changed] > 100).any()         # cv2 colours (0, 255, 0) and (255, 0, 0): B, G, R
    cells = read_xls(xl)["classification_results"]
    assert [cells[(r + 1, 0)] for r in range(len(listed))] == [os.path.basename(p) for p in listed]
    assert all(cells[(r + 1, 1)] in CLASS_LABELS for r in range(len(listed))) and (len(listed) + 1, 0) not in cells


def test_directory_drivers_show_the_reference_progress_bar_on_stderr(tmp_path, capsys, monkeypatch):
    """infer.py:46 / :79 walk the file list under tqdm: a progress bar on stderr, nothing on stdout.  Here: the same bar when tqdm
    is importable and stderr is a terminal, or ROOMNET_PROGRESS=1 asks for it (0 = never); it counts every file of the list."""
    pytest.importorskip("tqdm")
    from roomnet_amd import infer as I
    from roomnet_amd.imageio import imwrite
    rng = np.random.default_rng(2)
    paths = []
    for k in range(5):
        p = str(tmp_path / ("im%02d.png" % k))
        imwrite(p, rng.integers(0, 256, (30, 40, 3), dtype=np.uint8))
        paths.append(p)

    class Stub:
        im_side = 16

        def center_crop(self, x):
            h, w, _ = x.shape
            o = abs((w - h) // 2)
            return x[:, o:o + h, :] if h < w else (x[o:o + w, :, :] if w < h else x.copy())

        def infer(self, batch):
            return np.zeros(len(batch), np.int64), np.tile(np.float32([1, 0, 0, 0, 0, 0]), (len(batch), 1))

    monkeypatch.setenv("ROOMNET_PROGRESS", "1")
    assert len(list(I._infer_files(Stub(), paths, batch_size=2, decode_threads=2))) == 5
    cap = capsys.readouterr()
    assert "5/5" in cap.err and "5/5" not in cap.out
    monkeypatch.setenv("ROOMNET_PROGRESS", "0")
    assert len(list(I._infer_files(Stub(), paths, batch_size=2, decode_threads=2))) == 5
    assert "5/5" not in capsys.readouterr().err


def test_roomnet_keeps_the_reference_x_tensor_and_layers_attributes():
    """network.py:28-30, :207, :222: ``x_tensor`` (the input placeholder) and ``layers`` = [placeholder, one list per conv_block /
    dense_block].  Here they describe the same structure: 1 + 5 + 4 entries, the blocks' tensors by their tap names."""
    from roomnet_amd.network import RoomNet
    nn = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, optimized_inference=True)
    assert nn.x_tensor.shape == [None, 224, 224, 3] and nn.x_tensor.name == "input_x_tensor:0" and nn.x_tensor.dtype == np.float32
    assert nn.layers[0] is nn.x_tensor and len(nn.layers) == 10
    assert nn.layers[1] == ["s0.conv", "s0.pool", "s0.bn"]                       # network.py:226
    assert nn.layers[2][0] == "s1.conv" and nn.layers[2][-2:] == ["s3.add", "s3.bn2"] and len(nn.layers[2]) == 11   # :227, depth 3
    assert nn.layers[3][-1] == "s5.bn2" and nn.layers[4] == ["s6.conv", "s6.bn"] and nn.layers[5][-1] == "s9.bn2"   # :228-230
    assert nn.layers[6] == ["d0.mm", "d0.relu", "d0.bn"] and nn.layers[9] == ["d3.mm", "d3.relu"]                   # :234-237
    assert RoomNet(6, im_side=600, compute_bn_mean_var=False, optimized_inference=True).x_tensor.shape[1] == 600
