"""GPU tests of the drop-in Python surface (BASELINE config 1: the classify_im_dir driver on
8 x 224x224 images) against the oracle restating the same reference calls."""
import os

import numpy as np
import pytest

from conftest import MODEL_PREFIX
from oracle import c_oracle, roomnet_ref as R
from roomnet_amd import imageio
from roomnet_amd.imageops import resize_linear_u8
from xls_reader import read_xls

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nn():
    from roomnet_amd.network import RoomNet
    net = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, optimized_inference=True, max_batch=16)
    net.load(MODEL_PREFIX)
    yield net
    net.sess.close()


def test_infer_batch_matches_oracle(nn, weights, parity_images):
    ims = parity_images[[1, 9, 14, 22, 30, 38]]
    ids, probs = nn.infer(ims)
    ref = c_oracle.infer(weights, ims)
    assert ids.dtype == np.int64 and ids.shape == (6,) and probs.dtype == np.float32 and probs.shape == (6, 6)
    np.testing.assert_allclose(probs, ref["probs"], atol=1e-5, rtol=0)
    np.testing.assert_array_equal(ids, ref["ids"])
    # float input goes through the reference's float64 expression
    ids_f, probs_f = nn.infer(ims.astype(np.float64))
    np.testing.assert_array_equal(probs_f, probs)
    with pytest.raises(ValueError):
        nn.infer(ims[:, :200])


def test_infer_optimized_shapes_and_resize_path(nn, weights, parity_images):
    im = parity_images[22]
    idx, conf = nn.infer_optimized(im)
    assert idx.shape == (1,) and idx.dtype == np.int64 and conf.shape == (1, 6) and conf.dtype == np.float32
    ref = c_oracle.infer(weights, im[None])
    np.testing.assert_allclose(conf, ref["probs"], atol=1e-5, rtol=0)
    # non-square, non-224 input: centre crop + cv2-style bilinear resize on the host (network.py:149-152)
    rng = np.random.default_rng(11)
    big = (np.clip(np.add.outer(np.linspace(0, 200, 300), np.linspace(0, 55, 431))[..., None] +
                   rng.integers(0, 40, (300, 431, 3)), 0, 255)).astype(np.uint8)
    idx2, conf2 = nn.infer_optimized(big)
    cropped = R.center_crop(big)
    assert cropped.shape == (300, 300, 3)
    prepared = resize_linear_u8(np.ascontiguousarray(cropped), 224, 224)
    ref2 = c_oracle.infer(weights, prepared[None])
    np.testing.assert_allclose(conf2, ref2["probs"], atol=1e-5, rtol=0)
    assert idx2[0] == ref2["ids"][0]


def test_training_mode_infer_returns_ids_only(weights, parity_images):
    from roomnet_amd.network import RoomNet
    net = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, max_batch=4)
    net.load(MODEL_PREFIX)
    net.set_variables({k: v for k, v in weights.items() if k.startswith("dense") or "normalization_1" in k})
    out = net.infer(parity_images[[14, 30]])
    assert isinstance(out, np.ndarray) and out.dtype == np.int64 and out.shape == (2,)
    np.testing.assert_array_equal(out, c_oracle.infer(weights, parity_images[[14, 30]])["ids"])
    net.sess.close()


def test_save_then_load_none_round_trip_runs_the_same_bits(nn, weights, parity_images, tmp_path, monkeypatch, capsys):
    """SURVEY 8 f3 on the GPU: `save()` (network.py:93-103) -> a fresh model's `load(None)` (network.py:108-121: newest step in
    all_trained_models/trained_models) -> the engine built from the restored variables returns the bits of the original handle.
    Also the optimized-mode `save()` -> ./roomnet (network.py:94-97: how final_model/ was made) -> `load('roomnet')`."""
    from roomnet_amd.network import RoomNet
    monkeypatch.chdir(tmp_path)
    ims = parity_images[[1, 9, 14, 22, 30, 38]]
    ids0, probs0 = nn.infer(ims)
    s5_0 = nn._engine().tap("s5.bn2", len(ims))
    # a training-mode model holding the trained values (its restorer skips the dense blocks: network.py:242, so they are assigned)
    tr = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, max_batch=8)
    tr.load(MODEL_PREFIX)
    tr.set_variables({k: v for k, v in weights.items() if any(k.startswith(p) for p in tr._restore_excluded)})
    for step in (300, 4242, 77):              # load(None) must take the HIGHEST step, not the newest file
        tr.step = step
        if step == 4242:
            tr.save(suffix="0.88")
        else:
            w_other = {k: (v * 0.5).astype(np.float32) for k, v in weights.items()}
            from roomnet_amd import tf_bundle
            tf_bundle.write_bundle("all_trained_models/trained_models/roomnet--%d" % step, w_other)
    assert os.path.isfile("all_trained_models/trained_models/roomnet--0.88--4242.index")
    assert "Model saved at all_trained_models/trained_models/roomnet--0.88--4242" in capsys.readouterr().out
    np.testing.assert_array_equal(tr.infer(ims), ids0)
    tr.sess.close()
    fresh = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, optimized_inference=True, max_batch=8)
    fresh.load()                               # model_path None
    assert fresh.step == 4242 and fresh.start_step == 4242
    assert "Model restored from all_trained_models/trained_models/roomnet--0.88--4242" in capsys.readouterr().out
    ids1, probs1 = fresh.infer(ims)
    np.testing.assert_array_equal(probs1, probs0)
    np.testing.assert_array_equal(ids1, ids0)
    np.testing.assert_array_equal(fresh._engine().tap("s5.bn2", len(ims)), s5_0)
    # optimized-mode save(): ./roomnet.{index,data-00000-of-00001}, byte-identical data file, same bits through the engine
    fresh.save()
    assert "Model Saved in optimized inference mode" in capsys.readouterr().out
    assert open("roomnet.data-00000-of-00001", "rb").read() == open(MODEL_PREFIX + ".data-00000-of-00001", "rb").read()
    fresh.sess.close()
    again = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, optimized_inference=True, max_batch=8, dtype="bf16")
    again.load("roomnet")
    ref16 = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, optimized_inference=True, max_batch=8, dtype="bf16")
    ref16.load(MODEL_PREFIX)
    a, b = again.infer(ims), ref16.infer(ims)
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[0], b[0])
    again.sess.close()
    ref16.sess.close()


def test_classify_im_dir_end_to_end(nn, weights, parity_images, tmp_path, capsys):
    from roomnet_amd.infer import CLASS_LABELS, classify_im_dir
    d = tmp_path / "images"
    d.mkdir()
    pick = [1, 9, 14, 17, 22, 27, 30, 38]
    for k, i in enumerate(pick):
        assert imageio.imwrite(str(d / ("img_%02d.png" % k)), parity_images[i])
    (d / "notes.txt").write_text("not an image")          # the reference would crash here; we skip it
    ref = c_oracle.infer(weights, parity_images[pick])
    xl = classify_im_dir(nn, str(d), overlay=True, batch_size=3)
    assert xl == str(d) + "_classified_results.xls" and os.path.isfile(xl)
    out = capsys.readouterr().out
    assert "Classifying images in" in out and "Beginning inference.." in out and "unreadable image" in out
    cells = read_xls(xl)["classification_results"]
    assert cells[(0, 0)] == "IMAGE_NAME" and cells[(0, 1)] == "PREDICTED_LABEL"
    rows = {cells[(r, 0)]: (cells[(r, 1)], float(cells[(r, 2)])) for r in {r for r, _ in cells} if r > 0}
    assert len(rows) == 8
    for k in range(8):
        name = "img_%02d.png" % k
        label, conf = rows[name]
        assert label == CLASS_LABELS[ref["ids"][k]]
        assert abs(conf - ref["probs"][k, ref["ids"][k]]) <= 1e-5
        assert (name + " ---> " + label) in out.replace(str(d) + os.sep, "")
        written = str(d) + "_classified" + os.sep + label + os.sep + name
        assert os.path.isfile(written)
        im = imageio.imread(written)
        assert im.shape == (224, 224, 3) and (im != parity_images[pick[k]]).any()     # overlay drawn
    for lab in CLASS_LABELS:
        assert os.path.isdir(str(d) + "_classified" + os.sep + lab)
    # overlay=False copies the files untouched
    d2 = tmp_path / "copyset"
    d2.mkdir()
    imageio.imwrite(str(d2 / "a.png"), parity_images[14])
    classify_im_dir(nn, str(d2), overlay=False)
    lab = CLASS_LABELS[c_oracle.infer(weights, parity_images[14:15])["ids"][0]]
    copied = str(d2) + "_classified" + os.sep + lab + os.sep + "a.png"
    assert open(copied, "rb").read() == open(str(d2 / "a.png"), "rb").read()


def test_bf16_model_through_the_python_surface(weights, parity_images, golden_parity):
    from roomnet_amd.network import RoomNet
    net = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, optimized_inference=True,
                  dtype="bf16", max_batch=8)
    net.load(MODEL_PREFIX)
    ids, probs = net.infer(parity_images[:8])
    safe = golden_parity["top2_margin"][:8] > 0.2
    np.testing.assert_array_equal(ids[safe], golden_parity["ids"][:8][safe])
    net.sess.close()


def test_16bit_model_takes_integral_feeds_and_refuses_fractional_ones(parity_images):
    """The reference accepts any numeric array (network.py:128-135).  A 16-bit model fuses the uint8 table into its
    first kernel: integral arrays in [0, 255] are the same feed, anything else is refused with the reason."""
    from roomnet_amd.network import RoomNet
    net = RoomNet(num_classes=6, im_side=224, compute_bn_mean_var=False, optimized_inference=True, max_batch=4, dtype="bf16")
    net.load(MODEL_PREFIX)
    try:
        ims = parity_images[[9, 30]]
        ids, probs = net.infer(ims)
        for other in (ims.astype(np.int32), ims.astype(np.float64)):
            ids2, probs2 = net.infer(other)
            np.testing.assert_array_equal(probs2, probs)
            np.testing.assert_array_equal(ids2, ids)
        with pytest.raises(ValueError, match="dtype='f32'"):
            net.infer(ims.astype(np.float32) + 0.25)
        with pytest.raises(ValueError):
            net.infer(ims.astype(np.int32) - 1)
    finally:
        net.sess.close()


def test_groundtruth_validation_from_a_list_file(nn, weights, parity_images, tmp_path, capsys):
    """The batched validation driver (reference infer.py:41-57 with train.py:146-152's metrics): a list file whose
    paths contain spaces -> predictions equal to the oracle's ids -> accuracy / precision / recall / f-score equal to
    sklearn on those ids."""
    from sklearn.metrics import accuracy_score, precision_recall_fscore_support
    from roomnet_amd.infer import groundtruth_validation, read_fpaths
    d = tmp_path / "val set with spaces"
    d.mkdir()
    pick = list(range(3, 35, 2))                               # 16 images
    ref_ids = c_oracle.infer(weights, parity_images[pick])["ids"]
    rng = np.random.default_rng(5)
    truth = [int(ref_ids[k]) if rng.random() < 0.7 else int(rng.integers(0, 6)) for k in range(len(pick))]
    lines = []
    for k, i in enumerate(pick):
        p = d / ("room %02d.png" % k)
        assert imageio.imwrite(str(p), parity_images[i])
        lines.append("%s %d" % (p, truth[k]))
    lst = tmp_path / "val_list.txt"
    lst.write_text("\n".join(lines) + "\n")
    paths, labels, n = read_fpaths(str(lst))
    assert n == 16 and labels == truth and all(" " in p for p in paths)
    stats = groundtruth_validation(nn, str(lst), batch_size=5)
    out = capsys.readouterr().out
    assert "Inferring Images..." in out and "accuracy" in out
    acc = accuracy_score(truth, ref_ids)
    prec, rec, fsc, _ = precision_recall_fscore_support(truth, ref_ids, zero_division=0)
    assert stats["accuracy"] == pytest.approx(float(acc))
    np.testing.assert_allclose(stats["precisions"], prec)
    np.testing.assert_allclose(stats["recalls"], rec)
    np.testing.assert_allclose(stats["f-scores"], fsc)


def _mixed_size_set(parity_images):
    """Images of the sizes a real directory holds, made from the parity set by Pillow resampling (content only: what the
    drivers must reproduce is the crop + resize of THESE pixels): landscape, portrait, 1080p, the exact-2x case, an
    up-scale, and one that needs neither crop nor resize."""
    from PIL import Image
    sizes = [(300, 400), (500, 375), (1080, 1920), (448, 448), (100, 150), (224, 224)]       # (h, w)
    pick = [1, 9, 14, 22, 30, 38]
    out = []
    for (h, w), i in zip(sizes, pick):
        im = Image.fromarray(parity_images[i]).resize((w, h), Image.BICUBIC)
        out.append(np.ascontiguousarray(np.asarray(im, dtype=np.uint8)))
    return out


def _expected_prepared(im):
    """centre crop + cv2.resize of one image by the scalar oracle (the exact-2x case is OpenCV's 2x2 box average)."""
    from oracle import cv_resize_ref
    c = np.ascontiguousarray(cv_resize_ref.center_crop(im))
    if c.shape[0] == 224:
        return c
    if c.shape[0] == 448:
        v = c.astype(np.int32)
        return ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    return cv_resize_ref.resize_linear_u8_scalar(c, 224, 224)


def test_directory_drivers_feed_mixed_size_images_through_the_gpu_pipeline(nn, weights, parity_images, tmp_path, capsys):
    """classify_im_dir and groundtruth_validation hand the decoded images to the GPU as they are (crop + resize there,
    decode on a thread pool): ids and confidences equal those of the scalar cv2.resize restatement + the C oracle."""
    from roomnet_amd.infer import CLASS_LABELS, classify_im_dir, groundtruth_validation
    ims = _mixed_size_set(parity_images)
    prepared = np.stack([_expected_prepared(im) for im in ims], 0)
    ref = c_oracle.infer(weights, prepared)
    # the model's own batched entry for images of any size
    ids, probs = nn.infer_images(ims)
    np.testing.assert_array_equal(ids, ref["ids"])
    np.testing.assert_allclose(probs, ref["probs"], atol=1e-5, rtol=0)
    d = tmp_path / "mixed"
    d.mkdir()
    for k, im in enumerate(ims):
        assert imageio.imwrite(str(d / ("m_%d.png" % k)), im)
    (d / "broken.jpg").write_bytes(b"\xff\xd8 not a jpeg")
    xl = classify_im_dir(nn, str(d), overlay=False, batch_size=4)
    out = capsys.readouterr().out
    assert "unreadable image" in out
    cells = read_xls(xl)["classification_results"]
    rows = {cells[(r, 0)]: (cells[(r, 1)], float(cells[(r, 2)])) for r in {r for r, _ in cells} if r > 0}
    assert len(rows) == len(ims)
    for k in range(len(ims)):
        label, conf = rows["m_%d.png" % k]
        assert label == CLASS_LABELS[ref["ids"][k]]
        assert abs(conf - ref["probs"][k, ref["ids"][k]]) <= 1e-5
        assert os.path.isfile(str(d) + "_classified" + os.sep + label + os.sep + "m_%d.png" % k)
    lst = tmp_path / "mixed_list.txt"
    lst.write_text("".join("%s %d\n" % (d / ("m_%d.png" % k), int(ref["ids"][k]) if k % 2 else 5) for k in range(len(ims))))
    stats = groundtruth_validation(nn, str(lst), batch_size=4)
    truth = [int(ref["ids"][k]) if k % 2 else 5 for k in range(len(ims))]
    assert stats["accuracy"] == pytest.approx(float(np.mean(np.array(truth) == ref["ids"])))
