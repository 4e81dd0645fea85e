#!/usr/bin/env python3
"""Diagnostic (not part of the product): LDS cycles of the strip kernels' ring reads under the gfx950 banking rules.

A wave64 ds_read_b128 is served in four groups of sixteen lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
(MI355X_MICROARCH.md, LDS) -- one LDS cycle per group when its sixteen 16-byte accesses fall into sixteen different bank quads
(bank = (address / 4) mod 64), one more for every further distinct address on a busy quad.  The model walks every tile and tap of the
benchmark geometry (22 x 22 x 9 voxels per sample, strips of 11 rows, records of 128 bytes, row pitch Tp = 11) for both lane maps
(32x32x16: lane = voxel, half-wave = chunk; 16x16x32: lane = (voxel & 15, chunk)) and both swizzle keys (of the record index: rounds
2-3a; of the unpadded voxel run: docs/notebook_r1-r5.md section 4.1f) and prints cycles per read and the conflict share
(conflict cycles / active cycles = rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: measured 0.62 before, 0.11-0.13 after)."""
import collections

Tp, To, Wt, Wp, REC, NS = 11, 9, 22, 24, 128, 4
nvr = Wt * To
G = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G += [[l + 32 for l in g] for g in G]


def cycles(addrs):
    tot = 0
    for g in G:
        cnt, seen = collections.Counter(), set()
        for l in g:
            if addrs[l] not in seen:
                seen.add(addrs[l])
                cnt[(addrs[l] // 16) % 16] += 1
        tot += max(cnt.values())
    return tot


def record(tile_vox, dh, dw, dt, NV):
    vi = min(tile_vox, NV - 1)
    hrel, rem = divmod(vi, nvr)
    w, t = divmod(rem, To)
    wp, tp = w + dw, t + dt
    return ((hrel + dh) % NS) * Wp * Tp + wp * Tp + tp, wp, tp


def run(form, key):
    NV = 11 * nvr
    tot = n = 0
    for tile in range((NV + 31) // 32):
        for dh in range(3):
            for dw in range(3):
                for st in range(6 if form == "32x32x16" else 6):
                    dt, sel = st >> 1, st & 1                    # sel: k-block (32x32x16) / voxel half u (16x16x32)
                    addrs = []
                    for lane in range(64):
                        if form == "32x32x16":
                            vox, chunk = tile * 32 + (lane & 31), 2 * sel + (lane >> 5)
                        else:
                            vox, chunk = tile * 32 + 16 * sel + (lane & 15), lane >> 4
                        r, wp, tp = record(vox, dh, dw, dt, NV)
                        x = wp * (Tp - 2) + tp
                        if key == "record":
                            pos = chunk ^ ((r >> 1) & 7)
                        elif form == "32x32x16":
                            pos = chunk ^ ((x >> 1) & 7)
                        else:
                            pos = chunk ^ (((x >> 1) & 3) << 1)
                        addrs.append(r * REC + pos * 16)
                    tot += cycles(addrs)
                    n += 1
    c = tot / n
    print(f"{form:9s} key of the {key:12s}: {c:5.2f} LDS cycles per ds_read_b128, conflict share {(c - 4) / c:.2f}")


if __name__ == "__main__":
    for form in ("32x32x16", "16x16x32"):
        for key in ("record", "voxel run"):
            run(form, key)
