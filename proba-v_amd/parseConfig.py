"""cfg-file reader with the behaviour of the reference's utils/parseConfig.py:5-82.

Format: INI-like text; `#` comments and blank lines skipped; `[Section]` headers; `key=value` lines.
Values are typed by section and by *substring* of the key, exactly as the reference does:

  [Preprocessing]  'ckpt' -> list[int]; 'low_res_patch_thresholds' -> list[float];
                   'low_res_threshold' / 'high_res_threshold' -> float; 'to_flip' / 'to_rotate' -> bool(int);
                   anything else -> int
  [Net]            'decay_rate' -> float; 'is_grayscale' -> bool(int); else int
  [Train]          'learning_rate' / 'split' -> float; 'optimizer' / 'loss' -> str; else int
  other sections   str

Keys outside the whitelist raise AssertionError for every section but the first (reference :62-74);
all sections are merged into one flat dict (reference :77-80).
"""
import os

_KNOWN = frozenset("""raw_data preprocessing_out model_out batch_size epochs learning_rate optimizer split
num_res_blocks num_low_res_imgs num_low_res_imgs_pre scale num_filters kernel_size exp_rate decay_rate
is_grayscale max_shift patch_size patch_stride low_res_patch_thresholds low_res_threshold high_res_threshold
num_low_res_permute to_flip to_rotate ckpt test_out train_out loss""".split())


def _has(key, *needles):
    return any(n in key for n in needles)


def _typed(section, key, raw):
    txt = raw.strip()
    if section == "Preprocessing":
        if _has(key, "ckpt"):
            return [int(t) for t in raw.split(",")]
        if _has(key, "low_res_patch_thresholds"):
            return [float(t) for t in raw.split(",")]
        if _has(key, "low_res_threshold", "high_res_threshold"):
            return float(txt)
        if _has(key, "to_flip", "to_rotate"):
            return bool(int(txt))
        return int(txt)
    if section == "Net":
        if _has(key, "decay_rate"):
            return float(txt)
        if _has(key, "is_grayscale"):
            return bool(int(txt))
        return int(txt)
    if section == "Train":
        if _has(key, "learning_rate", "split"):
            return float(txt)
        if _has(key, "optimizer", "loss"):
            return txt
        return int(txt)
    return txt


def parseConfig(path):
    """Parse a model cfg file into one flat dict (reference utils/parseConfig.py:5)."""
    if not path.endswith(".cfg"):
        path += ".cfg"
    if not os.path.exists(path) and os.path.exists(os.path.join("cfg", path)):
        path = os.path.join("cfg", path)

    sections = []                                   # [(name, {key: value})] in file order
    with open(path, "r") as fh:
        for raw in fh.read().split("\n"):
            if not raw or raw.startswith("#"):
                continue
            line = raw.strip()
            if line.startswith("["):
                sections.append((line[1:-1].strip(), {}))
                continue
            key, val = line.split("=")              # exactly one '=' per line, as in the reference
            name, body = sections[-1]
            body[key.strip()] = _typed(name, key.strip(), val)

    unknown = []
    for _, body in sections[1:]:                    # the first section is not checked (reference :68)
        unknown += [k for k in body if k not in _KNOWN and k not in unknown]
    assert not unknown, "Unsupported fields {} in {}".format(unknown, path)

    if not sections:
        raise KeyError("type")                      # the reference deletes the merged dict's 'type' entry (:79): a file without sections fails there
    config = {}
    for _, body in sections:
        config.update(body)
    return config
