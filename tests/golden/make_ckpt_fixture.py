"""Extracts the tensor table (key, dtype, shape, shard, offset, size) of the reference's own checkpoint index
`modelInfo/ckpt_p16t9c85r12/NIR/ckpt-124.index` into tests/golden/ckpt124_index.json.  The table is data (the weights
themselves live in a shard the reference repository does not ship); it pins the variable inventory the engine must
reproduce.   python tests/golden/make_ckpt_fixture.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from probav_amd.tfckpt import read_index      # noqa: E402

idx = read_index("/root/reference/modelInfo/ckpt_p16t9c85r12/NIR/ckpt-124")
out = {k: ({"num_shards": v["num_shards"]} if k == "" else {"dtype": v["dtype"], "shape": list(v["shape"]), "shard": v["shard"],
                                                          "offset": v["offset"], "size": v["size"]}) for k, v in idx.items()}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ckpt124_index.json"), "w") as fh:
    json.dump(out, fh, indent=0, sort_keys=True)
print(len(out), "entries")
