#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/.

Run in the build container (needs /root/reference for the two fixtures that are
derived from the reference's shipped artefacts):

  python tools/make_golden.py

Fixtures
  graph_nodes_224.json   op / output-shape of every inference node recorded in the
                         reference's final_model/roomnet.meta (`_output_shapes`),
                         plus the TF version string -- the reference's own shape pin.
  bundle_index.json      name / shape / offset / size / masked CRC32C of the 79
                         tensors in final_model/roomnet.index.
  class_fields.npz       (input of this script; written by tools/search_class_images.py)
                         24 coarse colour grids, 4 per class of infer.py:22, whose
                         up-sampled images the checkpoint classifies with fp64 top-2
                         margins of 2.4-3.8.
  parity_224.npz         SELF-GENERATED (TF parity unpinned): for the 40 seeded
                         images of roomnet_amd.synth.parity_batch(224, seed=1) followed
                         by the 24 class-covering field images (synth.parity_set):
                         fp64-truth logits/probs/ids, fp32 logits, top-2 margins.
  taps_224.npz           SELF-GENERATED: for image 14 of that batch, per graph node:
                         mean, abs-max and 16 sampled elements (fp64 truth).
  parity_600.npz         SELF-GENERATED: 16 images of synth.parity_set(600) with the
                         seeded synthetic dense/kernel, chosen by
                         fp64 top-2 margin (none below 0.25: no ties).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from roomnet_amd import tf_bundle  # noqa: E402
from roomnet_amd.synth import parity_set  # noqa: E402
from oracle import roomnet_ref as R  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference/final_model/roomnet"
TAP_IMAGE = 14
TAP_SAMPLES = 16


def _shape_of(buf):
    dims = []
    unknown = False
    for f, wt, v in tf_bundle._proto_fields(buf):
        if f == 2:
            size = 0
            for f2, _w, v2 in tf_bundle._proto_fields(v):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(int(size))
        elif f == 3 and v:
            unknown = True
    return None if unknown else dims


def probe_meta(path):
    buf = open(path, "rb").read()
    info = {}
    nodes = []
    for f, wt, v in tf_bundle._proto_fields(buf):
        if f == 1:  # MetaInfoDef
            for f2, _w, v2 in tf_bundle._proto_fields(v):
                if f2 == 5:
                    info["tensorflow_version"] = v2.decode()
                elif f2 == 6:
                    info["tensorflow_git_version"] = v2.decode()
        elif f == 2:  # GraphDef
            for f2, _w, v2 in tf_bundle._proto_fields(v):
                if f2 != 1:
                    continue
                node = {"name": "", "op": "", "inputs": [], "attrs": {}}
                for f3, _w3, v3 in tf_bundle._proto_fields(v2):
                    if f3 == 1:
                        node["name"] = v3.decode()
                    elif f3 == 2:
                        node["op"] = v3.decode()
                    elif f3 == 3:
                        node["inputs"].append(v3.decode())
                    elif f3 == 5:
                        key, val = None, None
                        for f4, _w4, v4 in tf_bundle._proto_fields(v3):
                            if f4 == 1:
                                key = v4.decode()
                            elif f4 == 2:
                                val = v4
                        node["attrs"][key] = val
                nodes.append(node)
    return info, nodes


def _attr_summary(node):
    out = {}
    a = node["attrs"]
    if "_output_shapes" in a:
        shapes = []
        for f, _w, v in tf_bundle._proto_fields(a["_output_shapes"]):
            if f == 1:  # ListValue
                for f2, _w2, v2 in tf_bundle._proto_fields(v):
                    if f2 == 7:
                        shapes.append(_shape_of(v2))
        out["output_shapes"] = shapes
    for key in ("strides", "ksize"):
        if key in a:
            vals = []
            for f, wt, v in tf_bundle._proto_fields(a[key]):
                if f == 1:
                    for f2, wt2, v2 in tf_bundle._proto_fields(v):
                        if f2 == 3:
                            if wt2 == 2:  # packed
                                p = 0
                                while p < len(v2):
                                    x, p = tf_bundle._get_varint(v2, p)
                                    vals.append(x)
                            else:
                                vals.append(v2)
            out[key] = vals
    for key in ("padding", "data_format"):
        if key in a:
            for f, _w, v in tf_bundle._proto_fields(a[key]):
                if f == 2:
                    out[key] = v.decode()
    for key in ("align_corners", "is_training"):
        if key in a:
            for f, _w, v in tf_bundle._proto_fields(a[key]):
                if f == 5:
                    out[key] = bool(v)
    if "epsilon" in a:
        import struct
        for f, _w, v in tf_bundle._proto_fields(a["epsilon"]):
            if f == 4:
                out["epsilon"] = struct.unpack("<f", struct.pack("<I", v))[0]
    return out


COMPUTE_OPS = {"Conv2D", "Relu6", "AvgPool", "FusedBatchNorm", "ResizeBilinear", "Add", "Reshape",
               "MatMul", "BiasAdd", "Softmax", "ArgMax", "Placeholder", "Mul", "Sub", "Rsqrt"}


def make_graph_nodes():
    info, nodes = probe_meta(REF + ".meta")
    keep = []
    for n in nodes:
        if n["op"] not in COMPUTE_OPS:
            continue
        if "/Initializer/" in n["name"] or n["name"].startswith("save/") or "learn_rate" in n["name"]:
            continue
        d = {"name": n["name"], "op": n["op"], "inputs": n["inputs"]}
        d.update(_attr_summary(n))
        keep.append(d)
    with open(os.path.join(GOLD, "graph_nodes_224.json"), "w") as f:
        json.dump({"source": "final_model/roomnet.meta of the reference (MetaGraphDef), "
                             "decoded by tools/make_golden.py",
                   "info": info, "nodes": keep}, f, indent=1)
    print("graph nodes:", len(keep), info)


def make_bundle_index():
    r = tf_bundle.BundleReader(REF)
    ents = []
    for k in r.keys():
        e = r.entries[k]
        ents.append({"name": k, "shape": list(e.shape), "offset": e.offset, "size": e.size,
                     "crc32c_masked": e.crc32c})
    with open(os.path.join(GOLD, "bundle_index.json"), "w") as f:
        json.dump({"source": "final_model/roomnet.index of the reference", "header": r.header,
                   "entries": ents}, f, indent=1)
    print("bundle entries:", len(ents))


def sample_positions(size, k=TAP_SAMPLES):
    """Deterministic sample positions shared by the generator and the tests."""
    rng = np.random.default_rng(size)
    return np.sort(rng.integers(0, size, k))


def make_parity():
    w = tf_bundle.BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
    fields = np.load(os.path.join(GOLD, "class_fields.npz"))
    ims = parity_set(224, fields["fields_u8"])
    wanted = np.concatenate([np.full(len(ims) - len(fields["wanted_ids"]), -1, np.int64), fields["wanted_ids"]])
    logits64, probs64, ids64, logits32 = [], [], [], []
    taps = None
    for i in range(0, len(ims), 8):
        chunk = ims[i:i + 8]
        want_taps = i <= TAP_IMAGE < i + 8
        r64 = R.infer(w, chunk, np.float64, taps=want_taps)
        r32 = R.infer(w, chunk, np.float32)
        logits64.append(r64["logits"])
        probs64.append(r64["probs"])
        ids64.append(r64["ids"])
        logits32.append(r32["logits"])
        if want_taps:
            taps = {k: v[TAP_IMAGE - i] for k, v in r64["taps"].items()}
        print("parity chunk", i, r64["ids"])
    logits64 = np.concatenate(logits64)
    srt = np.sort(logits64, axis=1)
    np.savez_compressed(os.path.join(GOLD, "parity_224.npz"),
                        note=np.array("self-generated by oracle/roomnet_ref.py (fp64); TF parity unpinned"),
                        logits_f64=logits64, probs_f64=np.concatenate(probs64).astype(np.float64),
                        ids=np.concatenate(ids64), logits_f32=np.concatenate(logits32),
                        top2_margin=srt[:, -1] - srt[:, -2], wanted_ids=wanted)
    got = np.concatenate(ids64)
    assert (got[wanted >= 0] == wanted[wanted >= 0]).all(), "a class-covering image is not in its class any more"
    out = {"note": np.array("self-generated by oracle/roomnet_ref.py (fp64); TF parity unpinned"),
           "image_index": np.array(TAP_IMAGE)}
    for name in R.node_names():
        v = np.asarray(taps[name], np.float64).ravel()
        out[name + "|shape"] = np.array(taps[name].shape)
        out[name + "|mean"] = np.array(v.mean())
        out[name + "|absmax"] = np.array(np.abs(v).max())
        out[name + "|samples"] = v[sample_positions(v.size)]
    np.savez_compressed(os.path.join(GOLD, "taps_224.npz"), **out)
    print("classes reached:", sorted(set(np.concatenate(ids64).tolist())))


def make_parity_600(n_keep=16, min_margin=0.25):
    from oracle import c_oracle
    w = tf_bundle.BundleReader(os.path.join(ROOT, "roomnet_amd", "final_model", "roomnet")).load_all()
    w = dict(w)
    w["dense/kernel"] = R.synth_dense_kernel_600()
    fields = np.load(os.path.join(GOLD, "class_fields.npz"))["fields_u8"]
    ims = parity_set(600, fields)
    # pre-selection with the (fast) C restatement: per class the images with the largest fp32 margins, round-robin over the
    # classes the synthetic head reaches, so that the 16 kept images spread over as many ids as it offers
    pre = [c_oracle.infer(w, ims[i:i + 8]) for i in range(0, len(ims), 8)]
    lg = np.concatenate([p["logits"] for p in pre])
    ids = np.concatenate([p["ids"] for p in pre])
    srt = np.sort(lg, axis=1)
    m32 = srt[:, -1] - srt[:, -2]
    by_class = {c: sorted(np.nonzero((ids == c) & (m32 > 2 * min_margin))[0].tolist(), key=lambda i: -m32[i]) for c in sorted(set(ids.tolist()))}
    pick = []
    while len(pick) < n_keep and any(by_class.values()):
        for c in list(by_class):
            if by_class[c] and len(pick) < n_keep:
                pick.append(by_class[c].pop(0))
    pick = np.array(sorted(pick))
    r64 = {"logits": [], "probs": [], "ids": []}
    r32 = []
    for i in range(0, len(pick), 2):
        a = R.infer(w, ims[pick[i:i + 2]], np.float64)
        for k in r64:
            r64[k].append(a[k])
        r32.append(R.infer(w, ims[pick[i:i + 2]], np.float32)["logits"])
        print("600 chunk", i, a["ids"], flush=True)
    r64 = {k: np.concatenate(v) for k, v in r64.items()}
    srt = np.sort(r64["logits"], axis=1)
    margin = srt[:, -1] - srt[:, -2]
    assert margin.min() > min_margin, margin
    np.savez_compressed(os.path.join(GOLD, "parity_600.npz"),
                        note=np.array("self-generated; synthetic dense/kernel (seed 600); images = synth.parity_set(600, "
                                      "class_fields.npz fields)[image_indices]; TF parity unpinned"),
                        image_indices=pick,
                        logits_f64=r64["logits"], probs_f64=r64["probs"].astype(np.float64), ids=r64["ids"],
                        logits_f32=np.concatenate(r32), top2_margin=margin)
    print("600:", pick.tolist(), r64["ids"].tolist(), np.round(margin, 3).tolist())


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if os.path.isfile(REF + ".meta"):
        make_graph_nodes()
        make_bundle_index()
    else:
        print("reference not present: skipping graph_nodes / bundle_index")
    make_parity()
    make_parity_600()
