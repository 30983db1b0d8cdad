"""GPU: mirrors of the reference's accelerator tests (src/toast/tests/accelerator.py):
``test_memory`` (:131-184: create / update_device / update_host / delete round trip for every
dtype) and ``test_data_stage`` (:186-351: Data.accel_create / update_device / update_host /
delete by requires()-style dictionaries over detdata, shared and global objects)."""
import numpy as np
import pytest

from toast_amd import accel
from toast_amd.data import SharedData, defaults
from toast_amd.pixels import PixelData, PixelDistribution
from toast_amd.sim import create_satellite_data

pytestmark = pytest.mark.gpu

TYPES = {"f64": np.float64, "f32": np.float32, "i64": np.int64, "i32": np.int32, "i16": np.int16, "i8": np.int8,
         "u64": np.uint64, "u32": np.uint32, "u16": np.uint16, "u8": np.uint8}


@pytest.fixture(scope="module", autouse=True)
def device():
    assert accel.accel_enabled()
    accel.accel_assign_device(1, 0, 1.0, False)


def test_memory():
    data = {k: np.ones(100, dtype=tp) for k, tp in TYPES.items()}
    check = {k: 2 * np.array(v) for k, v in data.items()}
    for buf in data.values():
        assert not accel.accel_data_present(buf)
    for buf in data.values():
        accel.accel_data_create(buf)
        accel.accel_data_update_device(buf)
    for buf in data.values():
        assert accel.accel_data_present(buf)
    with pytest.raises(RuntimeError, match="already present"):
        accel.accel_data_create(data["f64"])
    for buf in data.values():
        buf[:] *= 2
        accel.accel_data_update_device(buf)
        buf[:] = 0
    for k, buf in data.items():
        accel.accel_data_update_host(buf)
        np.testing.assert_array_equal(buf, check[k])
    # reset zeroes the device copy only
    accel.accel_data_reset(data["i32"])
    np.testing.assert_array_equal(data["i32"], check["i32"])
    accel.accel_data_update_host(data["i32"])
    assert not np.any(data["i32"])
    for buf in data.values():
        accel.accel_data_delete(buf)
        assert not accel.accel_data_present(buf)
    with pytest.raises(RuntimeError, match="not present"):
        accel.accel_data_update_host(data["f64"])


def test_data_stage():
    data = create_satellite_data(n_det=4, n_samp=300)
    data.lazy_host = False
    ob = data.obs[0]
    names = {"global": ["test_pix"], "meta": [], "detdata": [], "shared": [], "intervals": []}
    for itp, (tname, tp) in enumerate(TYPES.items()):
        for sname, sshape in (("1", ()), ("2", (2,))):
            name = f"{tname}_{sname}"
            ob.detdata.create(name, sample_shape=sshape, dtype=tp)
            ob.detdata[name].data[:] = itp + 1
            ob.shared[name] = SharedData((itp + 1) * np.ones((ob.n_local_samples,) + sshape, dtype=tp), name)
            names["detdata"].append(name)
            names["shared"].append(name)
    dist = PixelDistribution(n_pix=100, n_submap=10, local_submaps=[0, 2, 4, 6, 8])
    data["test_pix"] = PixelData(dist, np.float64, n_value=3)
    data["test_pix"].data[:] = 7.0
    data.accel_create(names)
    data.accel_update_device(names)
    # clearing the host buffers does not touch the device copies
    for name in names["detdata"]:
        ob.detdata[name].buffer[:] = 0
        ob.shared[name].data[:] = 0
    data["test_pix"].buffer[:] = 0
    for name in names["detdata"]:
        assert ob.detdata[name].accel_in_use() and ob.shared[name].accel_in_use()
    data.accel_update_host(names)
    for itp, (tname, tp) in enumerate(TYPES.items()):
        for sname in ("1", "2"):
            name = f"{tname}_{sname}"
            assert np.all(ob.detdata[name].data == itp + 1), name
            assert np.all(ob.shared[name].data == itp + 1), name
            assert not ob.detdata[name].accel_in_use()
    assert np.all(data["test_pix"].data == 7.0)
    data.accel_delete(names)
    for name in names["detdata"]:
        assert not ob.detdata[name].accel_exists() and not ob.shared[name].accel_exists()
    assert not data["test_pix"].accel_exists()
    assert defaults.det_data in ob.detdata   # untouched objects stay where they were


_EVICT_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from toast_amd import ops
from toast_amd.data import defaults
from toast_amd.ops.pipeline import Pipeline
from toast_amd.sim import create_satellite_data

data = create_satellite_data(n_det=4, n_samp=100000)
dp = ops.PointingDetectorSimple()
pix = ops.PixelsHealpix(detector_pointing=dp, nside=64, skip_quaternions=False)
sw = ops.StokesWeights(detector_pointing=dp, mode="IQU", skip_quaternions=False)
Pipeline(operators=[dp, pix, sw]).apply(data)          # quats 12.8 MB + pixels 3.2 MB + weights 9.6 MB stay resident
ob = data.obs[0]
resident = [k for k in (defaults.quats, defaults.pixels, defaults.weights) if ob.detdata[k].accel_in_use()]
Pipeline(operators=[ops.Copy(detdata=[(defaults.det_data, "copy")])]).apply(data)   # + 2 x 3.2 MB
after = [k for k in (defaults.quats, defaults.pixels, defaults.weights) if ob.detdata[k].accel_exists()]
chk = [float(np.sum(ob.detdata[k].data.astype(np.float64))) for k in (defaults.quats, defaults.pixels, defaults.weights)]
same = bool(np.array_equal(ob.detdata["copy"].data, ob.detdata[defaults.det_data].data))
print("RESULT", len(resident), len(after), same, *chk)
"""


def test_eviction_under_memory_cap(tmp_path):
    """A failed device allocation evicts lazily retained detector data (written back to the host
    first) and retries: run once unconstrained and once with the manager capped at 28 MB
    (TOAST_HIP_MEM_LIMIT_MB), same results."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "evict.py"
    script.write_text(_EVICT_SCRIPT)
    out = {}
    for limit in (None, "28"):
        env = dict(os.environ)
        env.pop("TOAST_HIP_MEM_LIMIT_MB", None)
        if limit:
            env["TOAST_HIP_MEM_LIMIT_MB"] = limit
        res = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, env=env, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("RESULT")][0].split()
        out[limit] = line[1:]
    free, capped = out[None], out["28"]
    assert free[0] == "3" and free[1] == "3" and free[2] == "True"          # nothing evicted without a cap
    assert capped[0] == "3" and int(capped[1]) < 3 and capped[2] == "True"  # something had to go
    assert free[3:] == capped[3:]                                           # and came back intact
