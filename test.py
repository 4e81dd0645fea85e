#!/usr/bin/env python3
"""Inference CLI with the reference's flags and flow (test.py:25-100): --cfg --band --totest.

Loads <preprocessing_out>/resolverDir/<TEST|TRAIN>patchesLR_<band>.npy ([sets, 64, T, 1, 22, 22]), restores the latest
checkpoint, resolves every image set patch-wise on the MI355X engine (all sets in micro-batches of --micro-batch patches; forward, clip to
[0, 2**16], round half to even and the 8 x 8 stitch into 384 x 384 stay on the device) and writes uint16 PNGs named imgsetNNNN.png, skipping the ids in
removedTrainSets<band>.txt exactly as the reference does.
"""
import argparse
import logging
import os

import numpy as np
import torch

from probav_amd.modelsTF import WDSRConv3D
from probav_amd.parseConfig import parseConfig
from probav_amd.pngio import imsave_uint16
from probav_amd.testClass import evaluate, evaluate_device
from probav_amd.trainClass import ModelTrainer

logging.basicConfig(format="%(asctime)s - %(message)s", level=logging.INFO)
logger = logging.getLogger("probav_amd")

BAND_STATS = {"NIR": (8075.2045, 3160.7272), "RED": (5266.2245, 3431.8614)}
FIRST_ID = {("TEST", "NIR"): 1306, ("TEST", "RED"): 1160, ("TRAIN", "NIR"): 594, ("TRAIN", "RED"): 0}    # test.py:79-90


def parser():
    p = argparse.ArgumentParser()
    p.add_argument("--cfg", default="cfg/FINAL.cfg", type=str)
    p.add_argument("--band", type=str, default="RED")
    p.add_argument("--totest", type=str, default="TEST")
    p.add_argument("--micro-batch", type=int, default=2048, help="patches per forward launch; 16 = the reference's resolveByBatch (test.py:125). "
                   "Samples are independent, so the images do not depend on it")
    p.add_argument("--reference-loop", action="store_true", help="launch every micro-batch of 16 patches on its own, as the reference's loop does "
                   "(test.py:125-134; 3x slower, same pixels); by default the micro-batches are coalesced into launch sets")
    return p.parse_args()


def main(config, opt):
    logger.info("[ INFO ] Loading data...")
    dataDir = os.path.join(config["preprocessing_out"], "resolverDir")
    patchLR = np.load(os.path.join(dataDir, "%spatchesLR_%s.npy" % (opt.totest, opt.band)), allow_pickle=True)
    patchLR = np.array(patchLR).transpose((0, 1, 4, 5, 2, 3))                    # -> [sets, 64, 22, 22, T, 1] (test.py:38)
    mean, std = BAND_STATS["NIR" if opt.band == "NIR" else "RED"]
    k = config["kernel_size"]
    model = WDSRConv3D(name="superResolutionNet", band=opt.band, mean=mean, std=std, maxShift=config["max_shift"]).build(
        scale=config["scale"], numFilters=config["num_filters"], kernelSize=(k, k, k), numResBlocks=config["num_res_blocks"],
        expRate=config["exp_rate"], decayRate=config["decay_rate"], numImgLR=config["num_low_res_imgs"],
        patchSizeLR=config["patch_size"], isGrayScale=config["is_grayscale"]).to("cuda")
    basename = os.path.basename(opt.cfg).split(".")[0]
    ckptDir = os.path.join(config["model_out"], "ckpt_%s" % basename, opt.band)
    ModelTrainer(model, None, None, None, ckptDir, os.path.join(config["model_out"], "logs_%s" % basename, opt.band))   # restores the latest checkpoint
    logger.info("[ INFO ] Generating predictions...")
    y_preds = (evaluate_device(model, patchLR, micro_batch=16, launch_batch=16) if opt.reference_loop
               else evaluate_device(model, patchLR, micro_batch=opt.micro_batch))

    band = opt.band.upper()
    toOmit = []
    if os.path.exists("removedTrainSets%s.txt" % band):
        with open("removedTrainSets%s.txt" % band) as fh:
            toOmit = [int(float(line.split("\n")[0])) for line in fh.readlines()]
    outDir = (config["test_out"] if opt.totest == "TEST" else config["train_out"]) + "_" + basename
    i = FIRST_ID[("TEST" if opt.totest == "TEST" else "TRAIN", "NIR" if band == "NIR" else "RED")]
    os.makedirs(outDir, exist_ok=True)
    logger.info("[ SAVE ] Saving predicted images to %s..." % outDir)
    for img in y_preds:
        while i in toOmit:
            i += 1
        imsave_uint16(os.path.join(outDir, "imgset%04d.png" % i), img[:, :, 0].astype(np.uint16))
        i += 1


if __name__ == "__main__":
    opt = parser()
    main(parseConfig(opt.cfg), opt)
