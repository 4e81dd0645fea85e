#!/usr/bin/env python3
"""Generates tests/golden/cfg_cases.json by running the REFERENCE's own cfg parser (utils/parseConfig.py, pure stdlib -- the one
reference module that imports without TensorFlow) in the build container on a set of cfg texts: the two cfg files the reference
ships and synthetic ones that walk every typing rule, the whitelist and the failure modes.  Each case stores the input text and
what the reference returned (or the exception type it raised).  The reference file itself never ships; tests/test_arch_config.py
replays the cases through probav_amd.parseConfig.

    python tests/golden/make_cfg_fixture.py            # needs /root/reference
"""
import importlib.util
import json
import os
import tempfile

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg_cases.json")

SYNTHETIC = {
    "typing_rules": "# comment\n[Directories]\nanything_goes=here\nmodel_out = out dir \n\n[Train]\nbatch_size= 8\nsplit=0.2\nloss= l2 \nlearning_rate=5e-4\n"
                    "optimizer=adam\nepochs=100\n[Net]\ndecay_rate=0.5\nscale=3\nis_grayscale=0\nkernel_size=3\n"
                    "[Preprocessing]\nlow_res_patch_thresholds=0.85,0.9\nhigh_res_threshold=0.85\nlow_res_threshold = 0.7\nto_rotate=1\nto_flip=0\nckpt=2,3\npatch_stride=16\n",
    "substring_matching": "[Directories]\nraw_data=x\n[Preprocessing]\nckpt= 1 , 2\nmax_shift=6\nnum_low_res_permute=19\n[Train]\nsplit=1\n",
    "later_section_overrides": "[Directories]\nmodel_out=a\n[Train]\nbatch_size=8\n[Net]\nscale=3\n[Train]\nbatch_size=16\n",
    "spaces_in_header_and_values": "[ Directories ]\n raw_data = /a b/c \n[ Net ]\n num_filters = 32 \n",
    "unknown_key_in_first_section_is_allowed": "[Directories]\nwhatever=1\n[Train]\nepochs=3\n",
    "unknown_key_later": "[Directories]\nraw_data=x\n[Train]\nbatch_sizes=8\n",
    "unknown_section_keys_are_strings_but_whitelisted": "[Directories]\nraw_data=x\n[Extra]\nscale= 3 \n",
    "unknown_section_unknown_key": "[Directories]\nraw_data=x\n[Extra]\nfoo=3\n",
    "int_field_given_float": "[Directories]\nraw_data=x\n[Net]\nscale=3.0\n",
    "two_equal_signs": "[Directories]\nraw_data=a=b\n",
    "indented_comment_is_not_a_comment": "[Directories]\nraw_data=x\n  # not skipped\n",
    "whitespace_only_line": "[Directories]\nraw_data=x\n   \n[Net]\nscale=3\n",
    "key_before_any_section": "raw_data=x\n[Net]\nscale=3\n",
    "bool_from_nonzero_int": "[Directories]\nraw_data=x\n[Net]\nis_grayscale=2\n[Preprocessing]\nto_flip=-1\n",
    "empty_file": "",
}


def main():
    spec = importlib.util.spec_from_file_location("ref_parseConfig", os.path.join(REF, "utils", "parseConfig.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    texts = {}
    for name in ("p16t9c85r12", "p16t12c85r12"):
        with open(os.path.join(REF, "cfg", name + ".cfg")) as fh:
            texts["shipped_" + name] = fh.read()
    texts.update(SYNTHETIC)
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for name, text in texts.items():
            path = os.path.join(tmp, name + ".cfg")
            with open(path, "w") as fh:
                fh.write(text)
            case = {"name": name, "text": text}
            try:
                case["result"] = mod.parseConfig(path)
            except Exception as exc:                                   # noqa: BLE001 -- the exception TYPE is part of the behaviour
                case["error"] = type(exc).__name__
            cases.append(case)
    with open(OUT, "w") as fh:
        json.dump({"generator": "tests/golden/make_cfg_fixture.py", "reference_module": "utils/parseConfig.py:5-82", "cases": cases}, fh, indent=1, sort_keys=True)
    print("wrote %d cases to %s (%d raise)" % (len(cases), OUT, sum("error" in c for c in cases)))


if __name__ == "__main__":
    main()
