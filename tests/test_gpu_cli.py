"""The two CLIs as a user of the reference would run them (train.py:35-113, test.py:34-100): numpy.ma dumps in the layout the reference's
preprocessing writes, a cfg file, `python train.py --cfg ... --band NIR` -> a checkpoint under <model_out>/ckpt_<cfg>/<band>,
`python test.py --cfg ... --band NIR` -> uint16 PNGs named from the band's first id, skipping the ids of removedTrainSets<BAND>.txt
(read from the working directory, as the reference does), pixels equal to `testClass.evaluate` on the restored model."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from probav_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CFG = """[Directories]
raw_data={d}/raw
preprocessing_out={d}/pre
model_out={d}/modelInfo
train_out={d}/trainout
test_out={d}/testout

[Train]
batch_size=1
epochs=1
learning_rate=0.0005
optimizer=nadam
loss=l1
split=0.2

[Net]
num_res_blocks=12
num_low_res_imgs=9
scale=3
num_filters=32
kernel_size=3
exp_rate=8
decay_rate=0.8
is_grayscale=1

[Preprocessing]
max_shift=6
patch_size=16
patch_stride=16
num_low_res_imgs_pre=9
low_res_patch_thresholds=0.85
low_res_threshold=0.3
high_res_threshold=0.85
num_low_res_permute=0
to_flip=0
to_rotate=0
ckpt=1,2,3,4,5
"""


def _run(args, cwd, **extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PROBAV_FORCE_DP"):
        env.pop(k, None)
    env.update(extra)
    out = subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    return out


def test_train_py_then_test_py_are_drop_in(dev, tmp_path):
    d = str(tmp_path)
    aug, res = os.path.join(d, "pre", "augmentedPatchesDir"), os.path.join(d, "pre", "resolverDir")
    os.makedirs(aug), os.makedirs(res)
    # the reference's ModelTrainer evaluates (and only then saves) every 1000 steps of an epoch (models/trainClass.py:25,110-122):
    # 1000 training patches at batch_size 1 reach that point once
    n, nval = 1000, 6
    x, hr, mask = synth.synth_batch(16, seed=5)
    rep = lambda a, k: np.concatenate([a] * (k // len(a) + 1))[:k]
    for tag, k in (("TRAIN", n), ("TRAINVAL", nval)):
        np.ma.masked_array(rep(x, k), mask=np.zeros(rep(x, k).shape, bool)).dump(os.path.join(aug, "%spatchesLR_NIR.npy" % tag))
        np.ma.masked_array(rep(hr, k), mask=~rep(mask, k).astype(bool)).dump(os.path.join(aug, "%spatchesHR_NIR.npy" % tag))      # mask: True = obscured
    sets = 3
    test_patches = synth.synth_batch(sets * 64, seed=6)[0].reshape(sets, 64, 22, 22, 9, 1)
    np.ma.masked_array(test_patches.transpose(0, 1, 4, 5, 2, 3), mask=np.zeros((sets, 64, 9, 1, 22, 22), bool)).dump(
        os.path.join(res, "TESTpatchesLR_NIR.npy"))                                             # [sets, 64, T, 1, 22, 22] (utils/dataGenerator.py:118-120)
    cfg = os.path.join(d, "mini.cfg")
    with open(cfg, "w") as fh:
        fh.write(CFG.format(d=d))
    with open(os.path.join(d, "removedTrainSetsNIR.txt"), "w") as fh:
        fh.write("1307\n1308.0\n")

    out = _run([os.path.join(ROOT, "train.py"), "--cfg", cfg, "--band", "NIR"], cwd=d)
    ck = os.path.join(d, "modelInfo", "ckpt_mini", "NIR")
    assert open(os.path.join(ck, "checkpoint.pt-index")).read().split() == ["ckpt-1.pt"], out.stderr[-2000:]
    assert "[ EPOCH 0/1 ] - [ STEP 1000/1000 ]" in out.stderr and "VAL INFO" in out.stderr and "[ SAVE ] Saving checkpoint..." in out.stderr
    assert os.path.exists(os.path.join(d, "modelInfo", "logs_mini", "NIR", "events.jsonl"))

    _run([os.path.join(ROOT, "test.py"), "--cfg", cfg, "--band", "NIR"], cwd=d)
    pngs = sorted(os.path.basename(p) for p in glob.glob(os.path.join(d, "testout_mini", "*.png")))
    assert pngs == ["imgset1306.png", "imgset1309.png", "imgset1310.png"], pngs                  # test.py:79-100: first id 1306, 1307 / 1308 omitted

    # pixels: the restored model through the reference-shaped host loop (test.py:103-134), cast like test.py:99
    from probav_amd import testClass
    from probav_amd.modelsTF import WDSRConv3D
    from probav_amd.pngio import imread_uint16
    from probav_amd.trainClass import ModelTrainer
    m = WDSRConv3D("superResolutionNet", "NIR", 8075.2045, 3160.7272, 6).build(3, 32, (3, 3, 3), 12, 8, 0.8, 9, 16, True).to(dev)
    tr = ModelTrainer(m, None, None, None, ck, os.path.join(d, "lg"))
    assert tr.step == 1000
    want = testClass.evaluate(m, test_patches, batch_size=16)
    for name, w in zip(pngs, want):
        got = imread_uint16(os.path.join(d, "testout_mini", name))
        np.testing.assert_array_equal(got, w[:, :, 0].astype(np.uint16))
    # the training CLI under the data-parallel step (a world-size-1 RCCL group): resumes from the checkpoint and trains on
    out2 = _run([os.path.join(ROOT, "train.py"), "--cfg", cfg, "--band", "NIR"], cwd=d, PROBAV_FORCE_DP="1", MASTER_PORT="29577")
    assert "Model restored from checkpoint at step 1000" in out2.stdout
    assert open(os.path.join(ck, "checkpoint.pt-index")).read().split() == ["ckpt-1.pt", "ckpt-2.pt"]
