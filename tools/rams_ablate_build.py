"""Diagnostic builds for tools/rams_ablate.sh: `python tools/rams_ablate_build.py abl 8 16 24` -> libinrhip_r3abl<N>.so
(-DR3_ABLATE=<N>), `... skew 0 28 100` -> libinrhip_skew<N>.so (-DR3_SKEW=<N>).  The files are git-ignored; delete them after."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_build", os.path.join(ROOT, "mri-super-resolution_amd", "_build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
kind, vals = sys.argv[1], sys.argv[2:]
for v in vals:
    if kind == "abl":
        defs, name = ("-DR3_ABLATE=" + v,), "libinrhip_r3abl%s.so" % v
    elif kind == "stamps":
        defs, name = ("-DR3_STAMPS=1",), "libinrhip_r3stamps%s.so" % v
    else:
        defs, name = ("-DR3_SKEW=" + v,), "libinrhip_skew%s.so" % v
    print(b.build_diagnostic(defines=defs, out=os.path.join(ROOT, "mri-super-resolution_amd", name)))
