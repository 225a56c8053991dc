"""``RoomNet`` -- drop-in for the reference's ``network.RoomNet`` inference surface
(reference ``network.py:19-244``), executing on an MI355X through
libroomnet_hip.so instead of a TensorFlow session.

Kept from the reference (same names, argument meaning, return shapes/dtypes):
``RoomNet(num_classes, im_side=600, ...)``, attributes ``num_classes``,
``im_side``, ``sess``; ``init()``, ``load(model_path=None)``, ``save(suffix=None)``,
``center_crop(x)``, ``infer(im_batch)``, ``infer_optimized(im)``.
Training (``train_step``, loss/optimizer graph, ``network.py:49-85,158-170``) is out
of scope and raises ``NotImplementedError``.

MI355X-only keyword arguments (not in the reference): ``device``, ``dtype``
("f32" | "bf16" | "f16"), ``max_batch``.
"""
from __future__ import annotations

import os
from glob import glob
from typing import Dict, Optional

import numpy as np

from . import tf_bundle
from ._capi import Engine
from .graph import Graph, build_graph
from .imageops import resize_linear_u8


class _Session:
    """Stand-in for the ``tf.Session`` the reference keeps in ``self.sess``: owns the
    variable values; the device engine is (re)built from them on demand."""

    def __init__(self, variables: Dict[str, np.ndarray]):
        self.variables = variables
        self.engine: Optional[Engine] = None

    def close(self) -> None:
        if self.engine is not None:
            self.engine.close()
            self.engine = None


def _initializer_values(graph: Graph, seed: int = 0) -> Dict[str, np.ndarray]:
    """What ``tf.global_variables_initializer`` produces for this graph
    (``network.py:87-91``): glorot-uniform kernels, zero bias, BN gamma=1, beta=0,
    moving_mean=0, moving_variance=1."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in graph.variable_shapes().items():
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel":
            if len(shape) == 4:
                fan_in, fan_out = shape[0] * shape[1] * shape[2], shape[0] * shape[1] * shape[3]
            else:
                fan_in, fan_out = shape
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            out[name] = rng.uniform(-lim, lim, shape).astype(np.float32)
        elif leaf in ("gamma", "moving_variance"):
            out[name] = np.ones(shape, np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


class _Placeholder:
    """What the reference keeps in ``self.x_tensor`` (``network.py:28``: ``tf.placeholder(tf.float32, [None, S, S, 3],
    name='input_x_tensor')``): there is no TensorFlow graph here, so it is a description -- ``name``, ``shape``, ``dtype`` -- and
    the first entry of ``RoomNet.layers``."""

    def __init__(self, im_side):
        self.name = 'input_x_tensor:0'
        self.shape = [None, im_side, im_side, 3]
        self.dtype = np.float32

    def __repr__(self):
        return "<placeholder %r shape=(?, %d, %d, 3) dtype=float32>" % (self.name, self.shape[1], self.shape[2])


def _layer_names(graph: Graph):
    """``RoomNet.layers`` of the reference (``network.py:30``, ``:207``, ``:222``): the placeholder, then one list per conv_block /
    dense_block with that block's tensors in creation order.  Here the entries are the NAMES the per-layer read-out answers to
    (``RoomNet.tap(name)`` -> ``rn_tap``): ``sK.conv`` (conv + ReLU6), ``sK.pool``, ``sK.bn``, ``sK.add``, ``sK.bn2`` for conv stage K,
    ``dK.mm``, ``dK.relu``, ``dK.bn`` for dense block K.  A conv_block of depth d is d consecutive stages; the block that a stage
    with a skip connection closes started at its skip stage."""
    blocks, cur = [], []
    for s in graph.stages:
        names = ["s%d.conv" % s.index] + (["s%d.pool" % s.index] if s.pool_k else []) + ["s%d.bn" % s.index]
        if s.residual:
            names += ["s%d.add" % s.index, "s%d.bn2" % s.index]
        cur.append((s, names))
    # group: a residual stage ends the block its skip stage began; other stages are blocks of their own unless inside such a span
    i, n = 0, len(cur)
    ends = {s.skip_stage: s.index for s, _ in cur if s.residual}
    while i < n:
        j = ends.get(cur[i][0].index, cur[i][0].index)
        blocks.append([nm for k in range(i, j + 1) for nm in cur[k][1]])
        i = j + 1
    for d in graph.dense:
        blocks.append(["d%d.mm" % d.index, "d%d.relu" % d.index] + (["d%d.bn" % d.index] if d.bn_name else []))
    return blocks


class RoomNet:

    def __init__(self, num_classes, im_side=600, compute_bn_mean_var=True, start_step=0, dropout_enabled=False,
                 learn_rate=1e-4, l2_regularizer_coeff=1e-2, num_steps=10000, dropout_rate=.2,
                 update_batchnorm_means_vars=True, optimized_inference=False, *, device=0, dtype="f32",
                 max_batch=64):
        self.num_classes = num_classes
        self.im_side = im_side
        self.compute_bn_mean_var = compute_bn_mean_var
        self.optimized_inference = optimized_inference
        self.start_step = start_step
        self.step = start_step
        self.learn_rate = learn_rate
        self.dropout_enabled = False if optimized_inference else dropout_enabled
        self.model_folder = 'all_trained_models/trained_models'
        self.model_fpath_prefix = self.model_folder + '/' + 'roomnet-'
        if compute_bn_mean_var:
            # training=True batch statistics (network.py:193) need the training graph
            raise NotImplementedError("compute_bn_mean_var=True (batch-statistics BN) belongs to the training "
                                      "path, which is out of scope; construct with compute_bn_mean_var=False")
        self.graph = build_graph(num_classes=num_classes, im_side=im_side)
        # network.py:28-30: the input placeholder and the per-block list of layer tensors -- here their descriptions / tap names
        self.x_tensor = _Placeholder(im_side)
        self.layers = [self.x_tensor] + _layer_names(self.graph)
        self.device = device
        self.dtype = dtype
        self.max_batch = max_batch
        self.sess: Optional[_Session] = None
        # variables created by the dense blocks are not restored in training mode
        # (restore_excluded_vars, network.py:242 / :78)
        self._restore_excluded = set()
        if not optimized_inference:
            for d in self.graph.dense:
                self._restore_excluded.add(d.name + "/")
                if d.bn_name:
                    self._restore_excluded.add(d.bn_name + "/")

    # ------------------------------------------------------------- persistence
    def init(self):
        """network.py:87-91."""
        if not self.sess:
            self.sess = _Session(_initializer_values(self.graph))

    def save(self, suffix=None):
        """network.py:93-103 (writes the index + data shard; no .meta graph)."""
        if not self.sess:
            self.init()
        if self.optimized_inference:
            tf_bundle.write_bundle('roomnet', self.sess.variables)
            print('Model Saved in optimized inference mode')
            return
        if suffix:
            save_fpath = self.model_fpath_prefix + '-' + suffix + '--' + str(self.step)
        else:
            save_fpath = self.model_fpath_prefix + '-' + str(self.step)
        tf_bundle.write_bundle(save_fpath, self.sess.variables)
        print('Model saved at', save_fpath)

    def load(self, model_path=None):
        """network.py:105-126."""
        if not self.sess:
            self.init()
        if model_path is None:
            if os.path.isdir(self.model_folder):
                existing_paths = glob(self.model_folder + '/*.index')
                if len(existing_paths) == 0:
                    print('No model found to restore from, initializing random weights')
                    return
                existing_ids = [int(p.split('--')[-1].replace('.index', '')) for p in existing_paths]
                selected_idx = np.argmax(existing_ids)
                self.step = existing_ids[selected_idx]
                self.start_step = self.step
                model_path = existing_paths[selected_idx].replace('.index', '')
            else:
                print('No model found to restore from, initializing random weights')
                return
        reader = tf_bundle.BundleReader(model_path)
        shapes = self.graph.variable_shapes()
        restored = dict(self.sess.variables)
        for name, shape in shapes.items():
            if any(name.startswith(p) for p in self._restore_excluded):
                continue
            if name not in reader:
                raise tf_bundle.BundleError("Key %s not found in checkpoint %r" % (name, model_path))
            val = reader.get(name)
            if tuple(val.shape) != tuple(shape):
                raise ValueError("Assign requires shapes of both tensors to match. lhs shape= %s rhs shape= %s "
                                 "(variable %s; is the checkpoint for im_side=%d?)"
                                 % (list(shape), list(val.shape), name, self.im_side))
            restored[name] = val.astype(np.float32)
        self.sess.close()
        self.sess.variables = restored
        print('Model restored from', model_path)

    def set_variables(self, values: Dict[str, np.ndarray]) -> None:
        """Assign variable values directly (what ``sess.run(tf.assign(...))`` does)."""
        if not self.sess:
            self.init()
        shapes = self.graph.variable_shapes()
        for k, v in values.items():
            if k not in shapes:
                raise KeyError("unknown variable %r" % k)
            if tuple(np.shape(v)) != tuple(shapes[k]):
                raise ValueError("variable %r: shape %s does not match %s" % (k, np.shape(v), shapes[k]))
            self.sess.variables[k] = np.asarray(v, np.float32)
        self.sess.close()

    def _engine(self) -> Engine:
        if not self.sess:
            raise RuntimeError("Attempted to use a closed Session. (call init() or load() first)")
        if self.sess.engine is None:
            self.sess.engine = Engine(self.graph, self.sess.variables, device=self.device, dtype=self.dtype,
                                      max_batch=self.max_batch)
        return self.sess.engine

    def tap(self, name, n=1):
        """The per-layer read-out the reference gets from ``sess.run(self.layers[k][j], ...)``: tensor ``name`` (an entry of
        ``self.layers``) of the LAST inference call, float32 ``[n, h, w, c]`` (``rn_tap``).  Float32 engines built with taps hold
        every node; the throughput engines hold the tensors their launches write (RoomNetLibraryError otherwise)."""
        return self._engine().tap(name, n)

    # ----------------------------------------------------------------- inference
    def infer(self, im_in):
        """network.py:128-135: [N,S,S,3] BGR batch already at im_side.  Returns
        ``(argmax int64[N], softmax float32[N,C])`` in optimized mode, ``argmax``
        alone otherwise (``outs_final`` differs: network.py:45 vs :72)."""
        im = np.asarray(im_in)
        if im.ndim != 4 or im.shape[1:] != (self.im_side, self.im_side, 3):
            raise ValueError("Cannot feed value of shape %s for Tensor 'input_x_tensor:0', which has shape "
                             "'(?, %d, %d, 3)'" % (im.shape, self.im_side, self.im_side))
        eng = self._engine()
        im = self._as_feed(im)
        if im.dtype == np.uint8:
            ids, probs = eng.forward_u8(im)
        else:
            # non-uint8 input: same float64 expression as the reference, cast at the feed
            x = (((im[:, :, :, [2, 1, 0]] / 255.) * 2) - 1).astype(np.float32)
            ids, probs = eng.forward_f32(x)
        if self.optimized_inference:
            return ids, probs
        return ids

    def _as_feed(self, im):
        """The reference takes any numeric array (network.py:128-135).  The 16-bit engines fuse the uint8 ->
        [-1, 1] table into their first kernel and only take uint8: integral-valued arrays in [0, 255] are cast
        (same values, same table), anything else is refused here with the reason instead of a library error."""
        if im.dtype == np.uint8 or self.dtype == "f32":
            return im
        if im.size and (np.issubdtype(im.dtype, np.integer) or bool(np.all(im == np.rint(im)))) \
                and im.min() >= 0 and im.max() <= 255:
            return im.astype(np.uint8)
        raise ValueError("RoomNet(dtype=%r) takes uint8 images (or integral values in [0, 255]); got %s with "
                         "non-integral or out-of-range values -- construct with dtype='f32' for float feeds"
                         % (self.dtype, im.dtype))

    def center_crop(self, x):
        """network.py:137-146."""
        h, w, _ = x.shape
        offset = abs((w - h) // 2)
        if h < w:
            x_pp = x[:, offset:offset + h, :]
        elif w < h:
            x_pp = x[offset:offset + w, :, :]
        else:
            x_pp = x.copy()
        return x_pp

    def infer_optimized(self, im_in):
        """network.py:148-156: one BGR HWC image of any size -> ``(idx[1], conf[1,C])``."""
        eng = self._engine()
        if isinstance(im_in, np.ndarray) and im_in.dtype == np.uint8 and im_in.ndim == 3 and im_in.shape[2] == 3:
            # crop + cv2.resize restatement on the GPU (rn_classify_images_u8): bit-identical to the host path below
            return eng.classify_images([im_in])
        im = self.center_crop(im_in)
        h, w, _ = im.shape
        if h != self.im_side or w != self.im_side:
            im = resize_linear_u8(im, self.im_side, self.im_side)
        im = self._as_feed(np.ascontiguousarray(im))
        if im.dtype == np.uint8:
            out_label_idx, out_label_conf = eng.forward_u8(im[None])
        else:
            x = (((im[:, :, [2, 1, 0]] / 255.) * 2) - 1).astype(np.float32)
            out_label_idx, out_label_conf = eng.forward_f32(x[None])
        return out_label_idx, out_label_conf

    def infer_images(self, images):
        """Batched ``infer_optimized``: a list of BGR HWC images of ANY size -> what ``infer`` returns for the batch of
        their centre-cropped, resized versions (network.py:149-152 per image, then network.py:128-135).  3-channel
        uint8 images -- what ``cv2.imread`` yields -- go to the GPU as they are: crop + INTER_LINEAR resize run there
        (``rn_classify_images_u8``, byte for byte the host restatement); anything else is prepared on the host.  Not
        in the reference (its caller loops over ``infer_optimized``, infer.py:79-82); the directory drivers use it."""
        images = list(images)
        if not images:
            empty = np.zeros((0,), np.int64), np.zeros((0, self.num_classes), np.float32)
            return empty if self.optimized_inference else empty[0]
        if all(isinstance(im, np.ndarray) and im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3 for im in images):
            ids, probs = self._engine().classify_images(images)
            return (ids, probs) if self.optimized_inference else ids
        prepared = []
        for im in images:
            im = self.center_crop(np.asarray(im))
            if im.shape[0] != self.im_side or im.shape[1] != self.im_side:
                im = resize_linear_u8(np.ascontiguousarray(im), self.im_side, self.im_side)
            prepared.append(np.ascontiguousarray(im))
        return self.infer(np.stack(prepared, 0))

    def train_step(self, x_in, y):
        raise NotImplementedError("training (network.py:158-170) is out of scope of the MI355X inference path")
