"""The C-ABI library loads, exports every symbol include/chordvis.h declares, and the record layouts of
the header, the numpy mirrors and the ctypes mirrors agree.  No device calls (runs without a GPU)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from chord_amd import records as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, "include", "chordvis.h")).read()
    declared = sorted(set(re.findall(r"\b(chordvis_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 40
    raw = C.CDLL(built_lib.LIB_PATH)
    missing = [n for n in declared if not hasattr(raw, n)]
    assert not missing, "declared in include/chordvis.h but not exported: %s" % missing
    assert set(declared) == set(built_lib.EXPORTED), "chord_amd/lib.py prototypes out of sync with the header"
    assert b"gfx950" in built_lib.lib.chordvis_version()
    # ... and NOTHING else: the library's C++ internals live in a namespace `chord` (the reference's own), kernels have host-side
    # handles; none of it may be a dynamic symbol of the drop-in (-fvisibility=hidden + csrc/chordvis.map)
    nm = subprocess.run(["nm", "-D", "--defined-only", built_lib.LIB_PATH], capture_output=True, text=True)
    assert nm.returncode == 0, nm.stderr
    exported = sorted(ln.split()[-1] for ln in nm.stdout.splitlines() if ln.strip())
    assert exported == declared, "dynamic symbols beyond the header's: %s" % sorted(set(exported) - set(declared))[:8]


def test_record_layouts_match_the_header():
    names = {"ChordMeshlet": R.MESHLET, "ChordMeshletGroup": R.MESHLET_GROUP, "ChordPrimitive": R.PRIMITIVE,
             "ChordMaterial": R.MATERIAL, "ChordObject": R.OBJECT, "ChordInstanceCullingView": R.INSTANCE_CULLING_VIEW,
             "ChordCameraView": R.CAMERA_VIEW, "ChordDrawCmd": R.DRAW_CMD}
    nested = {("ChordObject", f): "basicData." + f for f in
              ("localToTranslatedWorld", "translatedWorldToLocal", "localToTranslatedWorldLastFrame", "scaleExtractFromMatrix")}
    lines = []
    for cname, dt in names.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in dt.names:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, nested.get((cname, f), f)))
    from chord_amd import lib as L
    ctypes_mirrors = (("ChordAssetDesc", R.AssetDesc), ("ChordSceneDesc", R.SceneDesc), ("ChordHZBDesc", R.HZBDesc),
                      ("ChordStats", L.Stats), ("ChordLimits", L.Limits), ("ChordTileMarker", L.TileMarker),
                      ("ChordShadingTiles", L.ShadingTiles), ("ChordCountAndCmd", L.CountAndCmd), ("ChordHZB", L.HZB))
    for cname, ct in ctypes_mirrors:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in ct._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "chordvis.h"\nint main(void){\n%s\nreturn 0;}\n' % "\n".join(lines)
    with tempfile.TemporaryDirectory() as td:
        cpath, exe = os.path.join(td, "l.c"), os.path.join(td, "l")
        open(cpath, "w").write(src)
        cc = subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), cpath, "-o", exe], capture_output=True, text=True)
        assert cc.returncode == 0, cc.stderr[-1500:]
        out = dict(l.split() for l in subprocess.check_output([exe]).decode().splitlines())
    for cname, dt in names.items():
        assert int(out[cname]) == dt.itemsize, cname
        for f in dt.names:
            assert int(out["%s.%s" % (cname, f)]) == dt.fields[f][1], (cname, f)
    for cname, ct in ctypes_mirrors:
        assert int(out[cname]) == C.sizeof(ct), cname
        for f, _ in ct._fields_:
            assert int(out["%s.%s" % (cname, f)]) == getattr(ct, f).offset, (cname, f)
    # reference sizes (gltf.h:26-153, base.h:121-135,343-360)
    assert [int(out[n]) for n in ("ChordMeshlet", "ChordMeshletGroup", "ChordPrimitive", "ChordMaterial", "ChordObject",
                                  "ChordInstanceCullingView", "ChordDrawCmd")] == [64, 40, 96, 96, 224, 288, 12]


def test_hzb_desc_agrees_with_oracle(built_lib):
    import orc
    for w, h in ((3840, 2160), (1920, 1080), (256, 256), (160, 96), (641, 377), (64, 64), (4096, 4096)):
        a, b = built_lib.hzb_desc(w, h), orc.hzb_desc(w, h)
        assert bytes(a) == bytes(b), (w, h)


def test_device_entry_points_fail_loudly_without_a_gpu(built_lib):
    """No CPU fallback: without a device chordvis_create reports CHORDVIS_E_NO_DEVICE."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    ctx = C.c_void_p()
    assert built_lib.lib.chordvis_create(0, None, C.byref(ctx)) == built_lib.E_NO_DEVICE
    from chord_amd.renderer import VisibilityRenderer
    with pytest.raises(built_lib.ChordvisError):
        VisibilityRenderer(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "chord_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "import orc" not in text and "oracle/" not in text.replace("oracle/oracle", "oracle/"), \
                    "%s references the oracle" % f
