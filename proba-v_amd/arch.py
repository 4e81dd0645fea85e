"""Architecture tables of the reference's WDSR-B Conv3D network (host side, no arithmetic).

Layer inventory and order follow the reference's graph (models/modelsTF.py:15-203) and its own
checkpoints (modelInfo/ckpt_p16t9c85r12/NIR/ckpt-124.index: `model/layer_with_weights-K/{g,v,layer/bias}`,
K in Keras topological order).  The native library (csrc/engine.hip) builds the same table in C++;
tests assert the two agree (parameter count 535 267 for p16t9c85r12).

Flat parameter buffer: layers in checkpoint order, per layer  [ g (Cout) | v (taps*Cin*Cout) | bias (Cout) ],
`v` in Keras kernel layout [kh, kw, kt, Cin, Cout] (2-D layers [kh, kw, Cin, Cout]).
"""
from collections import namedtuple

Layer = namedtuple("Layer", "name vshape cout g_off v_off b_off")


def reducer_plan(numImgLR):
    """Per valid `convReducer_i`: (kernel size k, mirrored H/W pad, mirrored depth pad) -- models/modelsTF.py:62-69.

      9  -> ConvReduceAndUpscale   (:152-164)  3 reducers, mirror pad 1 on H,W before the first
      13 -> ConvReduceAndUpscalev3 (:123-150)  5 reducers, mirror pad 1 on H,W before the first three
      7  -> ConvReduceAndUpscalev2 (:166-175)  2 reducers, no pad
      19 -> ConvReduceAndUpscaleEx (:76-121, marked EXPERIMENTAL there)  10 reducers: 5x5x5 on a pad of 2 in H,W,T,
            then 3x3x3 with pads (2,2,1), (2,2,0), (2,2,0), (1,1,0), then five unpadded
    """
    a, b = (3, 1, 0), (3, 0, 0)
    plans = {9: (a, b, b), 13: (a, a, a, b, b), 7: (b, b),
             19: ((5, 2, 2), (3, 2, 1), (3, 2, 0), (3, 2, 0), a, b, b, b, b, b)}
    if numImgLR not in plans:
        raise ValueError("numImgLR=%r: the reference defines WDSRConv3D reducers for 7, 9, 13 and 19 only" % (numImgLR,))
    return plans[numImgLR]


def layer_table(numFilters=32, numResBlocks=12, expRate=8, decayRate=0.8, numImgLR=9, scale=3, inChannels=1):
    """Ordered list of Layer records and the total parameter count.  inChannels: 1 (isGrayScale=True) or 3 -- the two input-facing
    layers, mainConv1 and residConv1, are the only ones that see it (models/modelsTF.py:19-20, :23-27)."""
    f, s2 = numFilters, scale * scale
    dec = int(numFilters * decayRate)                      # models/modelsTF.py:182
    shapes = [("mainConv1", (3, 3, 3, inChannels, f))]
    for i in range(numResBlocks):
        shapes += [("expConv_%d" % i, (1, 1, 1, f, f * expRate)),
                   ("decConv_%d" % i, (1, 1, 1, f * expRate, dec)),
                   ("normConv_%d" % i, (3, 3, 3, dec, f))]
    shapes += [("convReducer_%d" % (i + 1), (k, k, k, f, f)) for i, (k, _, _) in enumerate(reducer_plan(numImgLR))]
    shapes += [("residConv1", (3, 3, inChannels, s2)), ("upscaleConv1", (3, 3, 3, f, s2)),
               ("residConv2", (3, 3, s2, s2)), ("residConv3", (3, 3, s2, s2))]
    layers, off = [], 0
    for name, vs in shapes:
        cout, nv = vs[-1], 1
        for d in vs:
            nv *= d
        layers.append(Layer(name, vs, cout, off, off + cout, off + cout + nv))
        off += 2 * cout + nv
    return layers, off
