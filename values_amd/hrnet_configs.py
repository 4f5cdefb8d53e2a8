"""MODEL.EXTRA dictionaries of the HRNet layouts the reference ships (configs/model/hrnet_config*.yaml: W48) and of the
public HRNet-W18 layout BASELINE config 4 names -- what a hydra config would hand to `HighResolutionNet(cfg)`
(hrnet_module.py:343-346)."""
from __future__ import annotations


def hrnet_w48_extra(dropout_final=True):
    """The shipped configs/model/hrnet_config*.yaml layout (W48)."""
    return {
        "DROPOUT_FINAL": dropout_final, "FINAL_CONV_KERNEL": 1,
        "STAGE1": {"NUM_MODULES": 1, "NUM_BRANCHES": 1, "BLOCK": "BOTTLENECK", "NUM_BLOCKS": [4], "NUM_CHANNELS": [64],
                   "FUSE_METHOD": "SUM"},
        "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "BLOCK": "BASIC", "NUM_BLOCKS": [4, 4], "NUM_CHANNELS": [48, 96],
                   "FUSE_METHOD": "SUM"},
        "STAGE3": {"NUM_MODULES": 4, "NUM_BRANCHES": 3, "BLOCK": "BASIC", "NUM_BLOCKS": [4, 4, 4],
                   "NUM_CHANNELS": [48, 96, 192], "FUSE_METHOD": "SUM"},
        "STAGE4": {"NUM_MODULES": 3, "NUM_BRANCHES": 4, "BLOCK": "BASIC", "NUM_BLOCKS": [4, 4, 4, 4],
                   "NUM_CHANNELS": [48, 96, 192, 384], "FUSE_METHOD": "SUM"},
    }


def hrnet_w18_extra(dropout_final=True):
    """The public HRNet-W18 layout (BASELINE config 4): widths 18/36/72/144, same blocks/modules as W48."""
    e = hrnet_w48_extra(dropout_final)
    e["STAGE2"]["NUM_CHANNELS"] = [18, 36]
    e["STAGE3"]["NUM_CHANNELS"] = [18, 36, 72]
    e["STAGE4"]["NUM_CHANNELS"] = [18, 36, 72, 144]
    return e
