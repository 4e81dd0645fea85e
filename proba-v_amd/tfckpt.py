"""Reader / writer for TensorFlow checkpoint bundles (`<prefix>.index` + `<prefix>.data-0000k-of-0000n`), the format
the reference saves with `tf.train.Checkpoint` / `CheckpointManager` (models/trainClass.py:33-39, test.py:58-67), so that
its trained weights load into the MI355X engine.  No TensorFlow needed.

Format (restated from the public LevelDB table format and tensorflow/core/protobuf/tensor_bundle.proto):
  * `.index` is a LevelDB SSTable: data blocks of prefix-compressed (key, value) entries + restart array, a 5-byte trailer per
    block (compression type, masked crc32c), a metaindex block, an index block, and a 48-byte footer ending in the magic
    0xdb4775248b80fb57.  TF writes it uncompressed.
  * key "" -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version}; every other key is a checkpoint key such as
    `model/layer_with_weights-3/v/.ATTRIBUTES/VARIABLE_VALUE` -> BundleEntryProto {1: dtype, 2: shape, 3: shard_id,
    4: offset, 5: size, 6: crc32c(fixed32)}.
  * tensor bytes sit raw (little-endian, C order) in the shard file at [offset, offset + size).

Object-graph keys of the reference's model: `model/layer_with_weights-K/{g, v, layer/bias, initialized}`,
K = 0..43 in Keras topological order == the order of `probav_amd.arch.layer_table`.
"""
import os
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}
_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"


# ---- varints / protobuf ---------------------------------------------------------------------------
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if b < 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _proto_fields(buf):
    """Yield (field_number, wire_type, value) of one protobuf message (value: int or bytes)."""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n]); pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _parse_entry(buf):
    e = {"dtype": 0, "shape": (), "shard": 0, "offset": 0, "size": 0, "crc32c": None}
    for fn, wt, v in _proto_fields(buf):
        if fn == 1:
            e["dtype"] = v
        elif fn == 2:
            dims = []
            for f2, _, v2 in _proto_fields(v):
                if f2 == 2:                                   # TensorShapeProto.Dim
                    size = 0
                    for f3, _, v3 in _proto_fields(v2):
                        if f3 == 1:
                            size = v3
                    dims.append(size)
            e["shape"] = tuple(dims)
        elif fn == 3:
            e["shard"] = v
        elif fn == 4:
            e["offset"] = v
        elif fn == 5:
            e["size"] = v
        elif fn == 6:
            e["crc32c"] = v
    return e


# ---- crc32c (Castagnoli), masked as LevelDB / TF store it -----------------------------------------------
def _make_table():
    tbl = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tbl.append(c)
    return tbl


_CRC_TABLE = _make_table()


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    for b in bytes(data):
        crc = _CRC_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- SSTable ---------------------------------------------------------------------------------------
def _block_entries(block):
    """(key, value) pairs of one LevelDB block (restart array at the end is only needed for seeking)."""
    nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _read_block(buf, offset, size):
    ctype = buf[offset + size]
    if ctype != 0:
        raise ValueError("compressed SSTable block (type %d): TensorFlow writes checkpoint indexes uncompressed" % ctype)
    return buf[offset:offset + size]


def read_index(prefix):
    """{checkpoint key: {"dtype", "shape", "shard", "offset", "size", "crc32c"}}, plus key "" -> {"num_shards": n}."""
    path = prefix if prefix.endswith(".index") else prefix + ".index"
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != _MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad SSTable magic)" % path)
    footer = buf[-48:]
    pos = 0
    _, pos = _varint(footer, pos); _, pos = _varint(footer, pos)            # metaindex handle
    ioff, pos = _varint(footer, pos); isize, pos = _varint(footer, pos)    # index handle
    out = {}
    for _, handle in _block_entries(_read_block(buf, ioff, isize)):
        boff, p2 = _varint(handle, 0)
        bsize, _ = _varint(handle, p2)
        for key, val in _block_entries(_read_block(buf, boff, bsize)):
            if key == b"":
                hdr = {fn: v for fn, _, v in _proto_fields(val)}
                out[""] = {"num_shards": hdr.get(1, 1)}
            else:
                out[key.decode()] = _parse_entry(val)
    return out


def shard_path(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix[:-6] if prefix.endswith(".index") else prefix, shard, num_shards)


def read_tensor(prefix, index, key):
    e = index[key]
    path = shard_path(prefix, e["shard"], index[""]["num_shards"])
    if not os.path.exists(path):
        raise FileNotFoundError("checkpoint shard %s is missing (the reference repository lists its weight shards as missing "
                                "large blobs: .MISSING_LARGE_BLOBS)" % path)
    with open(path, "rb") as fh:
        fh.seek(e["offset"])
        raw = fh.read(e["size"])
    if e["crc32c"] is not None and _mask(crc32c(raw)) != e["crc32c"]:
        raise ValueError("crc32c mismatch for %s" % key)
    return np.frombuffer(raw, dtype=_DTYPES[e["dtype"]]).reshape(e["shape"]).copy()


def model_variable_keys(num_layers):
    """[(K, 'g'|'v'|'bias', checkpoint key)] for the weight-normalised layers, K in checkpoint order."""
    out = []
    for k in range(num_layers):
        base = "model/layer_with_weights-%d/" % k
        out += [(k, "g", base + "g" + _SUFFIX), (k, "v", base + "v" + _SUFFIX), (k, "bias", base + "layer/bias" + _SUFFIX)]
    return out


def load_reference_checkpoint(model, prefix):
    """Load `model/layer_with_weights-K/{g,v,layer/bias}` of a reference checkpoint into a WDSRModel; returns the
    checkpoint's training step (`step` variable) if present.  Shapes are verified against the model's layer table."""
    index = read_index(prefix)
    params = {}
    for (k, name, key), L in zip(model_variable_keys(len(model.layers)), [L for L in model.layers for _ in range(3)]):
        if key not in index:
            raise KeyError("checkpoint %s has no %s" % (prefix, key))
        want = {"g": (L.cout,), "v": tuple(L.vshape), "bias": (L.cout,)}[name]
        if tuple(index[key]["shape"]) != want:
            raise ValueError("%s: checkpoint shape %s, model expects %s (%s)" % (key, index[key]["shape"], want, L.name))
        params.setdefault(L.name, {})[name] = read_tensor(prefix, index, key)
    model.load_variables(params)
    step_key = "step" + _SUFFIX
    return int(read_tensor(prefix, index, step_key).reshape(-1)[0]) if step_key in index else None


def load_reference_optimizer(model, prefix):
    """The Keras optimizer state a reference checkpoint carries next to the weights: the two slots
    `<variable>/.OPTIMIZER_SLOT/optimizer/{m,v}` of each of the 132 trainables, `optimizer/iter` and Nadam's running product
    `optimizer/momentum_cache`, plus the trainer's `psnr`.  Returns {"iter", "momentum_cache", "m", "v", "psnr"} with m / v as flat
    float32 arrays in the engine's parameter order, or None if the bundle has no slots (weights-only export)."""
    index = read_index(prefix)
    layers = model.layers
    total = layers[-1].b_off + layers[-1].cout
    m, v = np.zeros(total, np.float32), np.zeros(total, np.float32)
    for (k, name, key), L in zip(model_variable_keys(len(layers)), [L for L in layers for _ in range(3)]):
        base = key[:-len(_SUFFIX)] + "/.OPTIMIZER_SLOT/optimizer/"
        if base + "m" + _SUFFIX not in index:
            return None
        lo, hi = {"g": (L.g_off, L.v_off), "v": (L.v_off, L.b_off), "bias": (L.b_off, L.b_off + L.cout)}[name]
        m[lo:hi] = read_tensor(prefix, index, base + "m" + _SUFFIX).reshape(-1)
        v[lo:hi] = read_tensor(prefix, index, base + "v" + _SUFFIX).reshape(-1)
    get = lambda key, default: (read_tensor(prefix, index, key + _SUFFIX).item() if key + _SUFFIX in index else default)
    return {"iter": int(get("optimizer/iter", 0)), "momentum_cache": float(get("optimizer/momentum_cache", 1.0)),
            "m": m, "v": v, "psnr": float(get("psnr", 1.0))}


# ---- writer (single shard, one data block per ~4 KB, uncompressed) ---------------------------------------
def _block(entries, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (key, val) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(key), len(last)) and key[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val)) + key[shared:] + val
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _entry_proto(dtype_id, shape, shard, offset, size, crc):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(s) for s in shape))
    msg = b"\x08" + _put_varint(dtype_id) + b"\x12" + _put_varint(len(dims)) + dims
    if shard:
        msg += b"\x18" + _put_varint(shard)
    if offset:
        msg += b"\x20" + _put_varint(offset)
    msg += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return msg


def write_bundle(prefix, tensors):
    """Write {checkpoint key: ndarray} as a one-shard TensorFlow bundle readable by `tf.train.load_checkpoint`."""
    keys = sorted(tensors)
    data, entries = bytearray(), [(b"", b"\x08\x01\x1a\x02\x08\x01")]      # header: num_shards=1, version{producer=1}
    for k in keys:
        a = np.ascontiguousarray(tensors[k])
        raw = a.tobytes()
        entries.append((k.encode(), _entry_proto(_DTYPE_IDS[a.dtype], a.shape, 0, len(data), len(raw), _mask(crc32c(raw)))))
        data += raw
    with open(shard_path(prefix, 0, 1), "wb") as fh:
        fh.write(bytes(data))
    out, index_entries = bytearray(), []

    def emit(block):
        off = len(out)
        out.extend(block)
        out.extend(b"\x00" + struct.pack("<I", _mask(crc32c(block + b"\x00"))))
        return off, len(block)

    chunk = []
    for e in entries:
        chunk.append(e)
        if sum(len(k) + len(v) for k, v in chunk) > 4096:
            off, size = emit(_block(chunk))
            index_entries.append((chunk[-1][0], _put_varint(off) + _put_varint(size)))
            chunk = []
    if chunk:
        off, size = emit(_block(chunk))
        index_entries.append((chunk[-1][0], _put_varint(off) + _put_varint(size)))
    moff, msize = emit(_block([]))
    ioff, isize = emit(_block(index_entries, restart_interval=1))
    footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    with open(prefix + ".index", "wb") as fh:
        fh.write(bytes(out))


def save_reference_checkpoint(model, prefix, step=0, psnr=1.0, optimizer=None):
    """Export a WDSRModel in the reference's object-graph key scheme (model variables, `step`, `psnr`).
    optimizer: optional {"iter", "momentum_cache", "m", "v"} (flat m / v in the engine's parameter order) written as the Keras
    Nadam slots and counters, so that the reference -- or `load_reference_optimizer` -- resumes with its moments."""
    tensors = {"step" + _SUFFIX: np.array(step, np.int32), "psnr" + _SUFFIX: np.array(psnr, np.float32)}
    tv = [t.detach().cpu().numpy() for t in model.trainable_variables]
    lay3 = [L for L in model.layers for _ in range(3)]
    for i, (k, name, key) in enumerate(model_variable_keys(len(model.layers))):
        tensors[key] = tv[i]
        if name == "g":
            tensors["model/layer_with_weights-%d/initialized%s" % (k, _SUFFIX)] = np.array(True)
        if optimizer is not None:
            L = lay3[i]
            lo, hi = {"g": (L.g_off, L.v_off), "v": (L.v_off, L.b_off), "bias": (L.b_off, L.b_off + L.cout)}[name]
            base = key[:-len(_SUFFIX)] + "/.OPTIMIZER_SLOT/optimizer/"
            for slot in ("m", "v"):
                tensors[base + slot + _SUFFIX] = np.asarray(optimizer[slot], np.float32)[lo:hi].reshape(tv[i].shape)
    if optimizer is not None:
        tensors["optimizer/iter" + _SUFFIX] = np.array(int(optimizer["iter"]), np.int64)
        tensors["optimizer/momentum_cache" + _SUFFIX] = np.array(optimizer["momentum_cache"], np.float32)
    write_bundle(prefix, tensors)
