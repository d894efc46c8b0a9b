#!/usr/bin/env python3
"""Generates tests/golden/cascade_setup.json: inputs of cascadeComputeCS (cascade_setup.hlsl:79-372) and the float32 BIT
PATTERNS of its outputs as tests/spec_np.py cascade_views_f32 computes them -- a float-by-float numpy restatement of the shader
in source order (no FMA, pow in binary64 rounded once, round-half-even).  chordvis_cascade_setup is held to these bits with no
tolerance (tests/test_depth_views.py).  Pure numpy: runs anywhere; re-run only when the canonical arithmetic is redefined."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import spec_np as S
from chord_amd import lib as L, records as R, scenes

cases = []
cam = scenes.Camera((1.5, 2.0, 4.0), (0.2, -0.3, -1.0), 1280, 720)
view, iv = L.make_views(cam)                 # (the view block itself is pinned by glm_camera.json)
light = (0.35, -0.8, 0.45)
rng = [int(np.float32(0.0004).view(np.uint32)), int(np.float32(0.02).view(np.uint32))]
for name, kw, vr, tick, cache in (
        ("five_cascades_two_realtime", dict(cascadeCount=5, realtimeCascadeCount=2, cascadeDim=1024, cascadeEndDistance=30.0, farCascadeEndDistance=120.0), None, 0, False),
        ("reference_defaults_with_sdsm_range", dict(), rng, 0, False),
        ("reference_defaults_cached_tick_7", dict(), rng, 7, True)):
    cfg = R.default_cascade_config(**kw)
    out = S.cascade_views_f32(cfg, view, light, valid_range=None if vr is None else np.asarray(vr, np.uint32), tick=tick, cache_valid=cache)
    def bits(a): return [int(x) for x in np.asarray(a, np.float32).reshape(-1).view(np.uint32)]
    cases.append(dict(name=name, config={k: (float(cfg[k][0]) if cfg[k].dtype.kind == "f" else int(cfg[k][0])) for k in cfg.dtype.names},
                      zNear=float(view["zNear"][0]), zFar=float(view["zFar"][0]),
                      clipToTranslatedWorldWithZFar_NoJitter_bits=bits(view["clipToTranslatedWorldWithZFar_NoJitter"][0]),
                      lightDir=list(light), validDepthMinMax=vr, tick=tick, cacheValid=cache,
                      cascades=[None if o is None else dict(translatedWorldToClip_rc=bits(o["translatedWorldToClip"]), clipToTranslatedWorld_rc=bits(o["clipToTranslatedWorld"]),
                                                              planes=bits(o["planes"]), orthoDepthConvertToView=bits(o["ortho"])) for o in out]))
json.dump(dict(generator="tests/golden/make_cascade_fixture.py (tests/spec_np.py cascade_views_f32)", cases=cases), open(os.path.join(HERE, "cascade_setup.json"), "w"), indent=1)
print("wrote cascade_setup.json:", [(c["name"], sum(x is not None for x in c["cascades"])) for c in cases])
