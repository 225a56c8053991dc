"""Static description of the RoomNet inference graph.

Replays the reference's graph builder (``network.py:225-244`` ``init_nn_graph``
calling ``conv_block`` ``network.py:172-208`` and ``dense_block``
``network.py:210-223``) into a flat list of *stages*.  One stage is what one
fused HIP kernel executes::

    conv3x3 VALID, stride 1, no bias -> ReLU6 -> [avg-pool k x k, stride s, VALID]
        -> BN(inference) -> [ + legacy-bilinear-resize(skip) -> BN(inference) ]

TensorFlow names its variables by creation order (``conv2d``, ``conv2d_1`` ...,
``batch_normalization``, ``batch_normalization_1`` ...); the same counters are
replayed here so that checkpoint tensor names line up with stages.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

BN_EPSILON = 1e-3  # tf.layers.batch_normalization default, network.py:193


def _suffix(base: str, i: int) -> str:
    return base if i == 0 else "%s_%d" % (base, i)


@dataclass
class ConvStage:
    index: int
    cin: int
    cout: int
    in_side: int
    conv_side: int          # in_side - 2
    pool_k: int             # 0 = no pooling
    pool_s: int
    out_side: int           # side after pooling (== conv_side when pool_k == 0)
    conv_name: str          # e.g. "conv2d_4"
    bn_name: str            # BN applied after the pool
    skip_stage: int = -1    # stage whose output is resized+added (-1: none)
    skip_side: int = 0
    bn2_name: str = ""      # BN applied after the residual add

    @property
    def residual(self) -> bool:
        return self.skip_stage >= 0

    @property
    def flops(self) -> int:
        return 2 * self.conv_side * self.conv_side * 9 * self.cin * self.cout


@dataclass
class DenseLayer:
    index: int
    nin: int
    nout: int
    name: str               # "dense", "dense_1", ...
    bn_name: str = ""       # "" -> no BN (last layer)
    biased: bool = False


@dataclass
class Graph:
    im_side: int
    num_classes: int
    stages: List[ConvStage] = field(default_factory=list)
    dense: List[DenseLayer] = field(default_factory=list)

    @property
    def flat_len(self) -> int:
        s = self.stages[-1]
        return s.out_side * s.out_side * s.cout

    def variable_shapes(self):
        """name -> shape of every checkpoint variable the graph restores."""
        out = {}
        for s in self.stages:
            out[s.conv_name + "/kernel"] = (3, 3, s.cin, s.cout)
            for bn in (s.bn_name, s.bn2_name):
                if bn:
                    for p in ("beta", "gamma", "moving_mean", "moving_variance"):
                        out["%s/%s" % (bn, p)] = (s.cout,)
        for d in self.dense:
            out[d.name + "/kernel"] = (d.nin, d.nout)
            if d.biased:
                out[d.name + "/bias"] = (d.nout,)
            if d.bn_name:
                for p in ("beta", "gamma", "moving_mean", "moving_variance"):
                    out["%s/%s" % (d.bn_name, p)] = (d.nout,)
        return out

    # ---- traffic / flop model of SURVEY.md section 8(d) -------------------
    def flops_per_image(self) -> int:
        return sum(s.flops for s in self.stages)

    def boundary_elements_per_image(self) -> int:
        """Stage-boundary traffic model: every stage reads its input once and
        writes its post-BN output once; residual stages re-read their skip
        tensor once; the head reads flat_len and writes num_classes."""
        n = 0
        for s in self.stages:
            n += s.in_side * s.in_side * s.cin
            n += s.out_side * s.out_side * s.cout
            if s.residual:
                n += s.skip_side * s.skip_side * s.cout
        n += self.flat_len + self.num_classes
        return n


class _Builder:
    def __init__(self, im_side: int):
        self.side = im_side
        self.ch = 3
        self.n_conv = 0
        self.n_bn = 0
        self.n_dense = 0
        self.stages: List[ConvStage] = []

    def conv_block(self, output_filters: int, pooling: bool = True, pool_ksize: int = 3,
                   pool_stride: int = 1, block_depth: int = 1) -> None:
        """reference network.py:172-208 (kernel 3, stride 1, VALID, BN on)."""
        make_residual = block_depth > 1
        first: Optional[ConvStage] = None
        for depth in range(block_depth):
            conv_side = self.side - 2
            if conv_side < 1:
                raise ValueError("im_side too small for the RoomNet graph")
            if pooling:
                if conv_side < pool_ksize:
                    raise ValueError("im_side too small for the RoomNet graph")
                out_side = (conv_side - pool_ksize) // pool_stride + 1
            else:
                out_side = conv_side
            st = ConvStage(index=len(self.stages), cin=self.ch, cout=output_filters,
                           in_side=self.side, conv_side=conv_side,
                           pool_k=pool_ksize if pooling else 0,
                           pool_s=pool_stride if pooling else 1,
                           out_side=out_side,
                           conv_name=_suffix("conv2d", self.n_conv),
                           bn_name=_suffix("batch_normalization", self.n_bn))
            self.n_conv += 1
            self.n_bn += 1
            self.stages.append(st)
            if depth == 0:
                first = st
            self.side, self.ch = out_side, output_filters
        if make_residual:
            last = self.stages[-1]
            last.skip_stage = first.index
            last.skip_side = first.out_side
            last.bn2_name = _suffix("batch_normalization", self.n_bn)
            self.n_bn += 1


def build_graph(num_classes: int = 6, im_side: int = 224) -> Graph:
    """reference network.py:225-237."""
    b = _Builder(im_side)
    b.conv_block(8)
    b.conv_block(32, pool_ksize=4, pool_stride=1, block_depth=3)
    b.conv_block(64, pool_ksize=4, pool_stride=2, block_depth=2)
    b.conv_block(128, pooling=False)
    b.conv_block(16, pool_ksize=4, pool_stride=2, block_depth=3)
    g = Graph(im_side=im_side, num_classes=num_classes, stages=b.stages)
    nin = g.flat_len
    n_bn = b.n_bn
    for i, nout in enumerate((32, 16, 8)):
        g.dense.append(DenseLayer(i, nin, nout, _suffix("dense", i),
                                  bn_name=_suffix("batch_normalization", n_bn)))
        n_bn += 1
        nin = nout
    g.dense.append(DenseLayer(3, nin, num_classes, _suffix("dense", 3), biased=True))
    return g
