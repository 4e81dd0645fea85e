#!/usr/bin/env python3
"""Reference-EXECUTED fixtures for the tensor plumbing around the kernels (build container only: reads /root/reference).

The reference's model code needs TensorFlow, which is not installable here; but four of its plumbing functions are pure numpy / torch
and run as they are once lifted out of their modules (whose imports of tensorflow / skimage would fail).  This script parses the
reference files with `ast`, compiles ONLY the named function / method / statement nodes -- no text of the reference is written
anywhere -- executes them on seeded inputs and stores inputs + outputs in `tests/golden/ref_plumbing.npz`:

  test.py:149-160               reconstruct_from_patches(images)                -> 8x8 (and 4x4) row-major stitch
  test.py:125-134               resolveByBatch(model, lr_batch, batch_size)     -> micro-batch boundaries incl. the remainder (stub `resolve`)
  models/testClass.py:31-39     Enhancer.reconstruct(self, patches)             -> the hard-coded 4x4 grid of 96-pixel blocks
  utils/dataGenerator.py:569-596 generatePatches / generatePatchesPerImgSet     -> the unfold of the padded frames
  utils/dataGenerator.py:108-121 the statements that pad (reflect, max_shift//2) and reshape around that call, as they stand in main()
  test.py:38                    the transpose test.py applies to the dump before the model sees it

Every pixel of an input carries a unique integer id (exact in float32), so an output IS its gather map.
    python tests/golden/make_ref_fixtures.py
"""
import ast
import os

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _tree(rel):
    with open(os.path.join(REF, rel)) as fh:
        return ast.parse(fh.read(), filename=rel)


# What is lifted out of the reference is EXECUTED here, and the reference is untrusted content: every node is pinned to the sha256 of
# its `ast.dump` as reviewed when this script was written.  A reference that has changed since (or another checkout) is refused, not run.
# Manual tool: nothing in tests/ or the build invokes it; run it by hand in the build container (no network) to regenerate the fixture.
PINS = {
    "test.py::reconstruct_from_patches+resolveByBatch": "85d9ccfb6a750c7ffec44b481657403dc43f35226087af634a69a6b73cdc9e77",
    "models/testClass.py::Enhancer.reconstruct": "00d8bbe12dbcc21b42fa4415127aa44d468e64db4f4f62f269e1cc1fbdb9674d",
    "utils/dataGenerator.py::generatePatches+generatePatchesPerImgSet": "baac2cd268a0e1d9222e58a9b0b8420fd59dfc9d4ea938fd32ff5254532e4d7b",
    "utils/dataGenerator.py:106-121": "b9bfbd9371eb844f7b90c1a191d796146a4ebb0439862384adb15d54d2d54556",
}


def _pinned(key, nodes):
    import hashlib
    import sys
    digest = hashlib.sha256("\n".join(ast.dump(n, include_attributes=False) for n in nodes).encode()).hexdigest()
    if "--print-pins" in sys.argv:
        print('    "%s": "%s",' % (key, digest))
        return
    if PINS.get(key) != digest:
        raise SystemExit("make_ref_fixtures: the reference code behind %r is not the reviewed one (sha256 %s, pinned %s): "
                         "read it, then update PINS" % (key, digest, PINS.get(key)))


def _functions(rel, names, namespace):
    """Compile the top-level functions `names` of a reference module into `namespace` (nothing else of the module runs)."""
    tree = _tree(rel)
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in nodes) == sorted(names), (rel, [n.name for n in nodes])
    _pinned(rel + "::" + "+".join(sorted(names)), nodes)
    exec(compile(ast.Module(body=nodes, type_ignores=[]), os.path.join(REF, rel), "exec"), namespace)
    return {n.name: (n.lineno, n.end_lineno) for n in nodes}


def _method(rel, cls, name, namespace):
    tree = _tree(rel)
    (c,) = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls]
    (m,) = [n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == name]
    _pinned(rel + "::" + cls + "." + name, [m])
    exec(compile(ast.Module(body=[m], type_ignores=[]), os.path.join(REF, rel), "exec"), namespace)
    return (m.lineno, m.end_lineno)


def _statements(rel, first, last):
    """The statement nodes of a reference file that lie entirely inside [first, last] and whose parent does not: compiled as a block."""
    tree = _tree(rel)
    out = []

    def walk(body):
        for n in body:
            if first <= n.lineno and n.end_lineno <= last:
                out.append(n)
            else:
                for field in ("body", "orelse", "finalbody"):
                    sub = getattr(n, field, None)
                    if isinstance(sub, list) and sub and isinstance(sub[0], ast.stmt):
                        walk(sub)
    walk(tree.body)
    _pinned("%s:%d-%d" % (rel, first, last), out)
    return compile(ast.Module(body=out, type_ignores=[]), os.path.join(REF, rel), "exec"), [(n.lineno, n.end_lineno) for n in out]


def main():
    import torch
    from tqdm import tqdm
    out, lines = {}, {}

    # -- test.py:149-160 / 125-134 ---------------------------------------------------------------------------------------------
    ns = {"np": np}
    lines.update({"test.py::" + k: v for k, v in _functions("test.py", ["reconstruct_from_patches", "resolveByBatch"], ns).items()})
    ids = np.arange(64 * 48 * 48, dtype=np.float64).reshape(64, 48, 48, 1)
    out["rec64_in_shape"] = np.array(ids.shape)
    out["rec64_out"] = ns["reconstruct_from_patches"](ids).astype(np.int32)
    ids16 = np.arange(16 * 96 * 96, dtype=np.float64).reshape(16, 96, 96, 1)
    out["rec16_out"] = ns["reconstruct_from_patches"](ids16).astype(np.int32)
    calls = []

    def stub_resolve(model, lr_batch):          # stands in for test.py:114-122 (TensorFlow); resolveByBatch only slices and concatenates
        calls.append(int(lr_batch.shape[0]))
        return model(lr_batch)
    ns["resolve"] = stub_resolve
    model = lambda b: np.asarray(b)[:, :2, :2, 0, :] * 2.0 + 1.0
    cases = [(37, 16), (32, 16), (7, 16), (40, 8), (16, 16), (1, 16)]
    out["rbb_cases"] = np.array(cases)
    for n, bs in cases:
        lr = np.arange(n * 3 * 3 * 2, dtype=np.float32).reshape(n, 3, 3, 2, 1)
        del calls[:]
        res = ns["resolveByBatch"](model, lr, batch_size=bs)
        out["rbb_%d_%d_calls" % (n, bs)] = np.array(calls)
        out["rbb_%d_%d_out" % (n, bs)] = res.astype(np.float32)
    del calls[:]
    ns["resolveByBatch"](model, np.zeros((37, 3, 3, 2, 1), np.float32))          # the default batch_size
    out["rbb_default_calls"] = np.array(calls)

    # -- models/testClass.py:31-39 -------------------------------------------------------------------------------------------------
    ns2 = {"np": np}
    lines["models/testClass.py::Enhancer.reconstruct"] = _method("models/testClass.py", "Enhancer", "reconstruct", ns2)
    out["enh_out"] = ns2["reconstruct"](None, ids16).astype(np.int32)

    # -- utils/dataGenerator.py:569-596 and the statements of main() around the call (108-121), then test.py:38 -----------------------
    ns3 = {"np": np, "torch": torch, "tqdm": lambda it, **kw: it}
    lines.update({"utils/dataGenerator.py::" + k: v for k, v in
                  _functions("utils/dataGenerator.py", ["generatePatches", "generatePatchesPerImgSet"], ns3).items()})
    code, spans = _statements("utils/dataGenerator.py", 106, 121)
    lines["utils/dataGenerator.py::main[pad+unfold+reshape]"] = (spans[0][0], spans[-1][1])
    sets, T, H = 2, 9, 64
    frames = np.arange(sets * T * H * H, dtype=np.float32).reshape(sets, T, 1, H, H)
    mask = np.zeros(frames.shape, bool)
    mask[0, 0, 0, 5, 7] = True                                # one masked pixel: numpy collapses an all-clear mask to `nomask` and the reference asserts on shapes
    ns3.update(config={"max_shift": 6, "patch_size": 16}, trmImgMskLRTest=np.ma.masked_array(frames, mask=mask))
    exec(code, ns3)
    patches = np.array(ns3["patchesLR"])                      # [sets, n*n, T, 1, 22, 22] as dumped to TESTpatchesLR_<band>.npy
    assert patches.shape == (sets, 16, T, 1, 22, 22), patches.shape
    (t38,) = [n for n in ast.walk(_tree("test.py")) if isinstance(n, ast.Assign) and n.lineno == 38]
    ns4 = {"patchLR": patches}
    exec(compile(ast.Module(body=[t38], type_ignores=[]), "test.py", "exec"), ns4)
    lines["test.py::main[transpose]"] = (38, 38)
    out["unfold_frames_shape"] = np.array(frames.shape)
    out["unfold_patches"] = ns4["patchLR"].astype(np.int32)     # [sets, n*n, 22, 22, T, 1]: what model(x) is fed
    # the same with the real geometry (128 x 128 -> 64 patches), one set, three frames: patch order at full width
    frames128 = np.arange(1 * 3 * 128 * 128, dtype=np.float32).reshape(1, 3, 1, 128, 128)
    mask128 = np.zeros(frames128.shape, bool)
    mask128[0, 0, 0, 5, 7] = True
    ns3.update(trmImgMskLRTest=np.ma.masked_array(frames128, mask=mask128))
    exec(code, ns3)
    ns4 = {"patchLR": np.array(ns3["patchesLR"])}
    exec(compile(ast.Module(body=[t38], type_ignores=[]), "test.py", "exec"), ns4)
    out["unfold128_patches"] = ns4["patchLR"].astype(np.int32)

    # id maps are stored as first differences of the flattened array (+ shape): runs of +1 compress to almost nothing;
    # tests/test_ref_plumbing.py::_ids undoes it with a cumulative sum
    for k in ("rec64_out", "rec16_out", "enh_out", "unfold_patches", "unfold128_patches"):
        a = out.pop(k)
        out[k + "_shape"] = np.array(a.shape)
        out[k + "_diff"] = np.diff(a.ravel().astype(np.int64), prepend=0).astype(np.int32)
    out["reference_lines"] = np.array(["%s:%d-%d" % (k, a, b) for k, (a, b) in sorted(lines.items())])
    np.savez_compressed(os.path.join(HERE, "ref_plumbing.npz"), **out)
    for k in sorted(lines):
        print("executed %-55s lines %d-%d" % (k, *lines[k]))
    print("wrote", os.path.join(HERE, "ref_plumbing.npz"), os.path.getsize(os.path.join(HERE, "ref_plumbing.npz")), "bytes")


if __name__ == "__main__":
    main()
