"""Drop-in for the reference's inference driver ``infer.py`` (``infer.py:1-110``):
``CLASS_LABELS``, the module constants, ``force_makedir``, ``read_fpaths``,
``groundtruth_validation``, ``classify_im_dir`` and the ``__main__`` block, running on the
MI355X ``RoomNet`` of this package instead of TensorFlow.

``classify_im_dir(nn, imgs_dir, overlay=True)`` keeps the reference's observable behaviour:
one ``<imgs_dir>_classified/<label>/`` directory per class, each image written there with
the two overlay lines (or copied when ``overlay=False``), one printed line per image, and a
``<imgs_dir>_classified_results.xls`` workbook (sheet ``classification_results``, header
``IMAGE_NAME | PREDICTED_LABEL``, rows ``name | label | str(conf)``); it returns the
workbook path.  The reference classifies strictly one image per ``sess.run``; here decoded
images are grouped into batches for the GPU (``batch_size``), which does not change any
result because the graph has no cross-image coupling (inference-mode BN).
"""
from __future__ import annotations

import contextlib
from glob import glob
import os
import shutil
import sys

import numpy as np

from . import xls
from .imageio import imread, imwrite, put_text
from .imageops import resize_linear_u8
from .network import RoomNet

CLASS_LABELS = ['Backyard', 'Bathroom', 'Bedroom', 'Frontyard', 'Kitchen', 'LivingRoom']

INPUT_MODEL_PATH = './final_model/roomnet'
INPUT_IMAGES_DIR = './test_images/set2/images'
IMG_SIDE = 224

INPUT_IMG_PATH_LIST_FILE = 'val_list.txt'


def read_fpaths(list_fpath):
    """infer.py:31-38."""
    with open(list_fpath, 'r') as f:
        data = f.readlines()
    fpath_components = [fpath_set.strip().split(' ') for fpath_set in data]
    im_paths = [' '.join(fpath_component[:-1]) for fpath_component in fpath_components]
    class_id = [int(fpath_component[-1]) for fpath_component in fpath_components]
    n = len(class_id)
    return im_paths, class_id, n


def _prepare(nn, im):
    """The host half of ``RoomNet.infer_optimized`` (network.py:149-152); used for images the device pipeline does
    not take (anything but 3-channel uint8) and for model objects without ``infer_images``."""
    im = nn.center_crop(im)
    h, w, _ = im.shape
    if h != nn.im_side or w != nn.im_side:
        im = resize_linear_u8(np.ascontiguousarray(im), nn.im_side, nn.im_side)
    return np.ascontiguousarray(im)


def _classify(nn, ims):
    """``(ids, probs-or-None)`` for a list of decoded BGR images of any size.  ``RoomNet.infer_images`` hands the raw
    images to the GPU, which crops and resizes them (``rn_classify_images_u8``: byte for byte the host restatement of
    ``cv2.resize``); other model objects get the host-prepared batch through ``infer`` like the reference's caller."""
    if hasattr(nn, 'infer_images'):
        outs = nn.infer_images(ims)
    else:
        outs = nn.infer(np.stack([_prepare(nn, im) for im in ims], 0))
    return outs if isinstance(outs, tuple) else (outs, None)


class _Closer:
    def __init__(self, fn):
        self.close = fn


def _usable_cores():
    try:
        return len(os.sched_getaffinity(0))          # the cores THIS process may run on (cgroup / taskset aware)
    except (AttributeError, OSError):
        return os.cpu_count() or 1


# Pillow releases the GIL inside the decoder and imread's channel swap does too; past 16-32 threads the pool levels off on the
# page faults of its fresh 6-8 MB buffers (tools/bench_images.py --threads and tools/bench_decode.py sweep it: 16 is the optimum for
# VGA files, 32 is 13 % better for 1080p; DESIGN.md section 5 has the figures of the GPU box's 256-thread host).
DECODE_THREADS = max(1, min(16, _usable_cores()))


def _progress(n):
    """The reference walks its file list under ``tqdm(range(num_fpaths))`` (infer.py:46, :79): a progress bar on stderr.  Same here
    when tqdm is importable and stderr is a terminal or ROOMNET_PROGRESS=1 asks for it; returns an ``update(k)`` callable and a
    ``close()``."""
    try:
        from tqdm import tqdm
    except ImportError:
        return (lambda k=1: None), (lambda: None)
    want = os.environ.get('ROOMNET_PROGRESS')
    if want == '0' or (want is None and not sys.stderr.isatty()):
        return (lambda k=1: None), (lambda: None)
    bar = tqdm(total=n, file=sys.stderr)
    return bar.update, bar.close


def _infer_files(nn, fpaths, batch_size, decode_threads=None):
    """Yield ``(index, image_bgr, idx, conf)`` per readable file, in list order.  Files are decoded on a small thread
    pool (Pillow releases the GIL while it decodes) that runs up to two batches ahead of the GPU; the GPU gets the
    decoded images in batches of ``batch_size``."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    pending = []

    def flush():
        if not pending:
            return []
        ids, probs = _classify(nn, [p[1] for p in pending])
        # the confidence stays an np.float32 scalar like infer_outs[1][0][idx] of the reference (infer.py:84): its
        # printed form, round(conf * 100, 2) and str(conf) are float32 results
        res = [(p[0], p[1], int(ids[k]), (probs[k][ids[k]] if probs is not None else np.float32('nan')))
               for k, p in enumerate(pending)]
        pending.clear()
        return res

    nthreads = max(1, int(decode_threads or DECODE_THREADS))
    window = max(2 * batch_size, nthreads)
    todo = iter(enumerate(fpaths))
    inflight = deque()
    tick, done = _progress(len(fpaths))
    with contextlib.closing(_Closer(done)), ThreadPoolExecutor(max_workers=nthreads) as pool:
        def top_up():
            while len(inflight) < window:
                nxt = next(todo, None)
                if nxt is None:
                    return
                inflight.append((nxt[0], nxt[1], pool.submit(imread, nxt[1])))
        top_up()
        while inflight:
            i, fpath, fut = inflight.popleft()
            im = fut.result()
            tick(1)
            top_up()
            if im is None:
                # the reference crashes here (cv2.imread returns None, infer.py:81-82); report and go on
                print(fpath, '---> unreadable image, skipped')
                continue
            pending.append((i, im))
            if len(pending) >= batch_size:
                for r in flush():
                    yield r
        for r in flush():
            yield r


def groundtruth_validation(nn, list_fpath=None, batch_size=64):
    """infer.py:41-57, with the list file it reads made an argument (the reference's global is
    commented out, infer.py:28).  Prints and returns accuracy and per-class precision / recall /
    f-score, computed like ``train.py:146-152``."""
    from sklearn.metrics import accuracy_score, precision_recall_fscore_support
    fpaths, labels, num_fpaths = read_fpaths(list_fpath or INPUT_IMG_PATH_LIST_FILE)
    print('Inferring Images...')
    y_preds, y_truths = [], []
    for i, _im, idx, _conf in _infer_files(nn, fpaths, batch_size):
        y_preds.append(idx)
        y_truths.append(labels[i])
    acc = accuracy_score(y_truths, y_preds)
    prec, rec, fsc, supp = precision_recall_fscore_support(y_truths, y_preds, zero_division=0)
    performance_stats = {'accuracy': float(acc),
                         'precisions': list(map(float, list(prec))),
                         'recalls': list(map(float, list(rec))),
                         'f-scores': list(map(float, list(fsc)))}
    print(performance_stats)
    return performance_stats


def force_makedir(dir):
    """infer.py:60-62."""
    if not os.path.isdir(dir):
        os.makedirs(dir)


def _overlay_and_write(im, pred_label, pred_conf, out_fpath):
    """infer.py:87-93 for one image: the two overlay lines, then the file."""
    h, w, _ = im.shape
    put_text(im, "Predicted Class: " + pred_label, (int(.5 * w), int(.90 * h)), (h / 720.) * .85, (0, 255, 0))
    put_text(im, "Confidence: " + str(round(pred_conf * 100, 2)) + " %", (int(.5 * w), int(.95 * h)),
             (h / 720.) * .85, (255, 0, 0))
    return imwrite(out_fpath, im)


def classify_im_dir(nn, imgs_dir, overlay=True, batch_size=64):
    """infer.py:65-100.  The overlay and the encoding of the output file (the reference does both between two ``sess.run`` calls)
    run on a second thread pool behind the loop -- per 1920 x 1080 image they cost what decoding it cost -- and the function
    returns when every file is written; printed lines, workbook rows and file names are the loop's, in list order."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    print('Classifying images in', imgs_dir)
    all_im_paths = glob(imgs_dir + '/*')
    out_dir = imgs_dir + '_classified'
    xl_fpath = out_dir + '_results.xls'
    class_dirs = [out_dir + os.sep + CLASS_LABELS[i] for i in range(len(CLASS_LABELS))]
    for dir in class_dirs:
        force_makedir(dir)
    print('Beginning inference..')
    excel_file = xls.Workbook()
    sheet = excel_file.add_sheet('classification_results')
    sheet.write(0, 0, 'IMAGE_NAME')
    sheet.write(0, 1, 'PREDICTED_LABEL')
    row = 0      # unreadable files are skipped (the reference crashes on them): rows stay contiguous
    writers = ThreadPoolExecutor(max_workers=DECODE_THREADS) if overlay else None
    writing = deque()
    try:
        for i, im, idx, pred_conf in _infer_files(nn, all_im_paths, batch_size):
            fpath = all_im_paths[i]
            row += 1
            pred_label = CLASS_LABELS[idx]
            out_fpath_dir = out_dir + os.sep + pred_label
            print(fpath, '--->', pred_label, pred_conf)
            if overlay:
                writing.append(writers.submit(_overlay_and_write, im, pred_label, pred_conf,
                                              out_fpath_dir + os.sep + fpath.split(os.sep)[-1]))
                while len(writing) > 4 * DECODE_THREADS:      # (a bound on the decoded images held for the writers)
                    writing.popleft().result()
            else:
                shutil.copy(fpath, out_fpath_dir)
            sheet.write(row, 0, fpath.split(os.sep)[-1])
            sheet.write(row, 1, pred_label)
            sheet.write(row, 2, str(pred_conf))
        while writing:
            writing.popleft().result()
    finally:
        if writers is not None:
            writers.shutdown(wait=True)
    excel_file.save(xl_fpath)
    return xl_fpath


if __name__ == '__main__':
    nn = RoomNet(num_classes=len(CLASS_LABELS), im_side=IMG_SIDE, compute_bn_mean_var=False,
                 optimized_inference=True)
    nn.load(INPUT_MODEL_PATH)

    # stats = groundtruth_validation(nn)
    xl_out_path = classify_im_dir(nn, INPUT_IMAGES_DIR)
