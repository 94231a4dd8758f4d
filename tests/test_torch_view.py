"""
GPU: the zero-copy consumer API (procgen_amd/torch_view.py; SURVEY section 8 f4, reference hook procgen/env.py:132-135,145):
torch tensors that ALIAS the library's device buffers, for single-part handles and for joint / multi-device ones.
"""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import action_stream


def _make(n, game, **kw):
    from procgen_amd import ProcgenGym3Env

    kw.setdefault("rand_seed", 23)
    return ProcgenGym3Env(n, game, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("host_observations", [True, False])
def test_device_observations_alias_the_library_buffer_and_equal_the_landed_frames(host_observations):
    import torch
    from procgen_amd import torch_view
    from test_gpu_parity_at_scale import DeviceBuffers

    n = 48
    env = _make(n, "coinrun", extra_options={"host_observations": host_observations})
    ref = _make(n, "coinrun")  # lands its frames on the host, as the ABI has it
    t = torch_view.device_observations(env)
    assert t.dtype == torch.uint8 and tuple(t.shape) == (n, 64, 64, 3) and t.is_cuda and t.is_contiguous()
    # no copy: the tensor starts at the pointer the library reports, and stays the same object's memory over steps
    b = DeviceBuffers()
    env._lib.procgen_amd_device_buffers.argtypes = [C.c_void_p, C.POINTER(DeviceBuffers)]
    assert env._lib.procgen_amd_device_buffers(env._handle, C.byref(b)) == 0
    assert t.data_ptr() == b.ob and t.device.index == b.device_id
    views = torch_view.device_views(env)
    assert len(views) == 1 and views[0].ob.data_ptr() == b.ob and views[0].rew.data_ptr() == b.rew and views[0].game == "coinrun"
    acts = action_stream(n, 12, seed=4)
    for k in range(len(acts) + 1):
        rew, ob, first = env.observe()
        rrew, rob, rfirst = ref.observe()
        # the SAME tensor object shows every new step (it aliases the buffer the render kernel writes)
        assert np.array_equal(t.cpu().numpy(), rob["rgb"]), f"step {k}"
        assert np.array_equal(views[0].rew.cpu().numpy(), rrew) and np.array_equal(views[0].first.cpu().numpy().astype(bool), rfirst)
        assert np.array_equal(views[0].level_seed.cpu().numpy(), ref.info_arrays()["level_seed"])
        if host_observations:
            assert np.array_equal(ob["rgb"], rob["rgb"])
        if k < len(acts):
            env.act(acts[k])
            ref.act(acts[k])
    # a torch consumer on its own stream: a uint8 -> float conversion + mean per env, the first op of a policy
    x = t.float().mean(dim=(1, 2, 3))
    assert np.allclose(x.cpu().numpy(), rob["rgb"].reshape(n, -1).astype(np.float64).mean(axis=1), atol=1e-3)
    env.close()
    ref.close()


@pytest.mark.gpu
def test_views_of_a_joint_sharded_handle_cover_every_env(monkeypatch):
    import torch
    from procgen_amd import torch_view

    monkeypatch.setenv("PROCGEN_AMD_FAKE_DEVICES", "1")
    names = ["coinrun", "bigfish", "maze"]
    n = 96
    env = _make(n, ",".join(names), extra_options={"num_devices": 2})
    acts = action_stream(n, 6, seed=9)
    for a in acts:
        env.act(a)
    rew, ob, first = env.observe()
    views = torch_view.device_views(env)
    assert len(views) == 6
    seen = np.zeros(n, bool)
    for v in views:
        idx = v.global_indices()
        assert v.num_envs == n // 6 and v.env_stride == 3 and v.game == names[v.first_env % 3]
        assert np.array_equal(v.ob.cpu().numpy(), ob["rgb"][idx]) and np.array_equal(v.rew.cpu().numpy(), rew[idx])
        seen[idx] = True
    assert seen.all()
    with pytest.raises(ValueError):
        torch_view.device_observations(env)
    whole = torch_view.scatter_observations(env)
    assert tuple(whole.shape) == (n, 64, 64, 3) and np.array_equal(whole.cpu().numpy(), ob["rgb"])
    env.close()


def test_torch_view_imports_without_torch_being_touched():
    """the package's own import must not pull torch in (PyTorch is plumbing, imported by this one module when it is called)"""
    import subprocess
    import sys

    code = "import sys; sys.path.insert(0, %r); import procgen_amd, procgen_amd.torch_view; assert 'torch' not in sys.modules" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, "-c", code]).returncode == 0
