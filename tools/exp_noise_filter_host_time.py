# where the host time of ops.NoiseFilter goes at cfg-3 (no serialising trace): wraps the capi / fft entry points it calls
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["mapmaker_pcg.py", "--iter", "2"]
import importlib.util
spec = importlib.util.spec_from_file_location("wf", os.path.join(sys.path[0], "workflows", "mapmaker_pcg.py"))
wf = importlib.util.module_from_spec(spec)
from toast_amd import capi, fft as hipfft
from toast_amd.ops import noise_filter as nf
log = []
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); log.append((name, t0, time.perf_counter())); return r
    setattr(mod, name, g)
for name in ("accel_update_device_wait", "accel_update_device_finish"):
    wrap(capi, name)
# arrival times of the upload's parts: a thread polls toast_hip_accel_update_device_arrived
import threading
arrivals = []
_parts = capi.accel_update_device_parts
def parts_and_watch(buf, part_end, name="NA"):
    t0 = time.perf_counter()
    _parts(buf, part_end, name)
    log.append(("accel_update_device_parts", t0, time.perf_counter()))
    n = len(part_end)
    def watch():
        seen = set()
        while len(seen) < n:
            for k in range(n):
                if k not in seen:
                    try:
                        if capi.accel_update_device_arrived(buf, k):
                            seen.add(k); arrivals.append((name, k, time.perf_counter()))
                    except Exception:
                        seen.add(k)
            time.sleep(0.0005)
    threading.Thread(target=watch, daemon=True).start()
capi.accel_update_device_parts = parts_and_watch
for name in ("impulse_extents", "convolve_buffer", "extend_flags_buffer"):
    wrap(hipfft, name)
wrap(nf, "estimate_net_stack")
orig_exec = nf.NoiseFilter._exec
def timed_exec(self, *a, **k):
    t0 = time.perf_counter(); r = orig_exec(self, *a, **k)
    import torch; torch.cuda.synchronize(); t1 = time.perf_counter()
    print("NoiseFilter._exec %.1f ms" % (1e3 * (t1 - t0)))
    for name, a0, a1 in log:
        print("   %-28s start %7.1f  dur %7.1f ms" % (name, 1e3 * (a0 - t0), 1e3 * (a1 - a0)))
    for name, k, ta in arrivals:
        print("   part %d of %-20s arrived at %7.1f ms" % (k, name, 1e3 * (ta - t0)))
    return r
nf.NoiseFilter._exec = timed_exec
spec.loader.exec_module(wf)
wf.main()
