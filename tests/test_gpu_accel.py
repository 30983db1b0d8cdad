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
    data["test_pix"].raw[:] = 0
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
