#!/usr/bin/env python3
"""Training CLI with the reference's flags and flow (train.py:26-113): --cfg --band --modelType.

Loads the augmented patch pickles the reference's preprocessing writes
(<preprocessing_out>/augmentedPatchesDir/{TRAIN,TRAINVAL}patches{LR,HR}_<band>.npy, numpy.ma dumps), builds the
WDSR-B Conv3D network on the MI355X engine, and runs ModelTrainer.fitTrainData.  Under
`python -m torch.distributed.run --nproc-per-node N train.py ...` every rank trains on its shard of the data and the
flat gradient buffer is all-reduced once per step (RCCL over xGMI).
"""
import argparse
import logging
import os

import numpy as np
import torch

from probav_amd.loss import Losses
from probav_amd.modelsTF import WDSRConv3D
from probav_amd.parseConfig import parseConfig
from probav_amd.trainClass import ModelTrainer, make_optimizer

logging.basicConfig(format="%(asctime)s - %(message)s", level=logging.INFO)
logger = logging.getLogger("probav_amd")

BAND_STATS = {"NIR": (8075.2045, 3160.7272), "RED": (5266.2245, 3431.8614)}      # train.py:47-52


def parser():
    p = argparse.ArgumentParser()
    p.add_argument("--cfg", default="cfg/yourcfg.cfg", type=str)
    p.add_argument("--band", type=str, default="NIR")
    p.add_argument("--modelType", type=str, default="patchNet")
    return p.parse_args()


def patchNet(config, opt):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1 or os.environ.get("PROBAV_FORCE_DP") == "1":     # (PROBAV_FORCE_DP=1: the data-parallel step on ONE GPU, over a world-size-1 RCCL group)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=int(os.environ.get("RANK", "0")), world_size=world, device_id=torch.device("cuda", local))
    logger.info("[ INFO ] Loading data...")
    dataDir = os.path.join(config["preprocessing_out"], "augmentedPatchesDir")
    load = lambda n: np.load(os.path.join(dataDir, n % opt.band), allow_pickle=True)
    X_train, X_val = load("TRAINpatchesLR_%s.npy"), load("TRAINVALpatchesLR_%s.npy")
    y_train, y_val = load("TRAINpatchesHR_%s.npy"), load("TRAINVALpatchesHR_%s.npy")
    y_train_mask, y_val_mask = ~np.ma.getmaskarray(y_train), ~np.ma.getmaskarray(y_val)      # True = clear pixel
    mean, std = BAND_STATS["NIR" if opt.band == "NIR" else "RED"]
    X_train, X_val, y_train, y_val = (np.array(a) for a in (X_train, X_val, y_train, y_val))

    logger.info("[ INFO ] Building model...")
    k = config["kernel_size"]
    model = WDSRConv3D(name="superResolutionNet", band=opt.band, mean=mean, std=std, maxShift=config["max_shift"]).build(
        scale=config["scale"], numFilters=config["num_filters"], kernelSize=(k, k, k), numResBlocks=config["num_res_blocks"],
        expRate=config["exp_rate"], decayRate=config["decay_rate"], numImgLR=config["num_low_res_imgs"],
        patchSizeLR=config["patch_size"], isGrayScale=config["is_grayscale"], seed=0).to(torch.device("cuda", local))
    optimizer = make_optimizer(config["optimizer"], model, config["learning_rate"])
    target = config["scale"] * config["patch_size"]
    loss = Losses(targetShape=(target, target, 1))
    type_loss = {"l1": loss.shiftCompensatedL1Loss, "l2": loss.shiftCompensatedL2Loss,
                 "sobel_l1_mix": loss.shiftCompensatedL1EdgeLoss, "l1msssim": loss.shiftCompensatedRevSSIM}[config["loss"]]
    basename = os.path.basename(opt.cfg).split(".")[0]
    ckptDir = os.path.join(config["model_out"], "ckpt_%s" % basename, opt.band)
    logDir = os.path.join(config["model_out"], "logs_%s" % basename, opt.band)
    trainer = ModelTrainer(model=model, loss=type_loss, metric=loss.shiftCompensatedcPSNR, optimizer=optimizer,
                           ckptDir=ckptDir, logDir=logDir)
    trainer.fitTrainData(X_train, [y_train, y_train_mask], config["batch_size"], config["epochs"],
                         [X_val, y_val, y_val_mask], saveBestOnly=False, initEpoch=0)
    logger.info("[ SUCCESS ] Model checkpoint can be found in %s." % ckptDir)
    logger.info("[ SUCCESS ] Model logs can be found in %s." % logDir)


if __name__ == "__main__":
    opt = parser()
    config = parseConfig(opt.cfg)
    if opt.modelType != "patchNet":
        raise SystemExit("--modelType %s: only the patchNet (WDSR-B Conv3D) path is implemented; the reference's fusionNet "
                         "is a separate experimental model with hard-coded paths (train.py:116-188)" % opt.modelType)
    patchNet(config, opt)
