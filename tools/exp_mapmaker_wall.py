import os, sys, importlib.util
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
sys.argv = ["mapmaker_pcg.py"]
spec = importlib.util.spec_from_file_location("wf", os.path.join(root, "workflows", "mapmaker_pcg.py"))
wf = importlib.util.module_from_spec(spec); spec.loader.exec_module(wf)
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    wf.main()
st = wf.LAST_STATS
print("drain=%s MapMaker %.4f s  NoiseFilter %.4f  phases %s" % (os.environ.get("TOAST_HIP_DELETE_SYNC", "0"), st["mapmaker_s"], st["laps"]["NoiseFilter"], {k: round(v, 4) for k, v in st["phases_s"].items()}))
