"""State-trace wire format.  CPU: round trip, memory-mapped load, incremental writer, rejection of
foreign / truncated files.  GPU: a trace written from one run replays through a fresh env to bit-identical outputs."""
import numpy as np
import pytest
import torch

from isaaclab_amd.envs import state_trace as ST
from isaaclab_amd.envs.synthetic import Recording


def _recording(n=5, steps=3, dec=4, j=3, b=2):
    rec = Recording()
    rec.name, rec.num_envs, rec.decimation = "toy", n, dec
    shapes = {"root_transforms": (n, 7), "root_velocities": (n, 6), "dof_positions": (n, j), "dof_velocities": (n, j),
              "net_contact_forces": (n * b, 3)}
    off = 0
    for name, shp in shapes.items():
        size = int(np.prod(shp))
        rec.fields[name] = (off, size, shp)
        off += (size + 3) // 4 * 4
    g = torch.Generator().manual_seed(0)
    rec.frames = torch.randn(1 + steps * dec, off, generator=g)
    rec.actions = torch.randn(steps, n, j, generator=g)
    rec.env_origins = np.arange(n * 3, dtype=np.float32).reshape(n, 3)
    return rec


def test_round_trip_and_mmap(tmp_path):
    rec = _recording()
    path = str(tmp_path / "toy.mdptrace")
    ST.save_trace(path, rec)
    for mmap in (False, True):
        got = ST.load_trace(path, mmap=mmap)
        assert (got.name, got.num_envs, got.decimation) == ("toy", 5, 4) and got.fields == rec.fields
        np.testing.assert_array_equal(np.asarray(got.frames), rec.frames.numpy())
        np.testing.assert_array_equal(got.actions.numpy(), rec.actions.numpy())
        np.testing.assert_array_equal(got.env_origins, rec.env_origins)
        np.testing.assert_array_equal(got.field("dof_positions", 2), rec.field("dof_positions", 2)) if not mmap else None
    header, off = ST.read_header(path)
    assert off % 64 == 0 and header["quaternion_order"] == "xyzw"


def test_incremental_writer_equals_bulk_save(tmp_path):
    rec = _recording()

    class View:
        def __init__(self, frame):
            self.frame = frame

    for name, getter in ST.TraceWriter.GETTERS.items():
        if name in rec.fields:
            o, s, shp = rec.fields[name]
            setattr(View, getter, (lambda o=o, s=s, shp=shp: lambda self: self.frame[o : o + s].view(shp))())
    path = str(tmp_path / "inc.mdptrace")
    w = ST.TraceWriter(path, "toy", 5, 4, rec.fields, rec.frames.shape[1], rec.env_origins)
    w.append_frame(View(rec.frames[0]))
    for k in range(3):
        for i in range(4):
            w.append_frame(View(rec.frames[1 + 4 * k + i]))
        w.append_actions(rec.actions[k])
    w.close()
    got = ST.load_trace(path)
    mask = torch.zeros(rec.frames.shape[1], dtype=torch.bool)      # padding words between fields are not recorded
    for o, s, _ in rec.fields.values():
        mask[o : o + s] = True
    np.testing.assert_array_equal(got.frames[:, mask].numpy(), rec.frames[:, mask].numpy())
    np.testing.assert_array_equal(got.actions.numpy(), rec.actions.numpy())


def test_rejects_foreign_and_truncated_files(tmp_path):
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"not a trace at all, just bytes")
    with pytest.raises(ValueError, match="bad magic"):
        ST.load_trace(str(bad))
    rec = _recording()
    path = tmp_path / "t.mdptrace"
    ST.save_trace(str(path), rec)
    data = path.read_bytes()
    (tmp_path / "cut.mdptrace").write_bytes(data[: len(data) - 40])
    with pytest.raises(ValueError, match="truncated"):
        ST.load_trace(str(tmp_path / "cut.mdptrace"))
    (tmp_path / "v9.mdptrace").write_bytes(data[:8] + (9).to_bytes(4, "little") + data[12:])
    with pytest.raises(ValueError, match="version"):
        ST.load_trace(str(tmp_path / "v9.mdptrace"))
    rec.actions = rec.actions[:2]
    with pytest.raises(ValueError, match="do not match"):
        ST.save_trace(str(tmp_path / "x.mdptrace"), rec)


@pytest.mark.gpu
def test_gpu_replay_of_a_saved_trace(device, tmp_path):
    from conftest import to_np
    from isaaclab_amd.envs.synthetic import make_velocity_env

    steps = 4
    env, rec = make_velocity_env("anymal_d_rough", num_envs=32, device=str(device), steps=steps, seed=71, small_terrain=True)
    path = str(tmp_path / "run.mdptrace")
    ST.save_trace(path, rec)
    outs = []
    for t in range(steps):
        rec.feed(t)
        obs, rew, term, trunc, _ = env.step(rec.actions[t].to(device))
        outs.append((to_np(obs["policy"]), to_np(rew), to_np(term), to_np(trunc)))
    # a fresh env of the same task, fed from the file (same injected random numbers: they belong to the run, not the trace)
    env2, rec2 = make_velocity_env("anymal_d_rough", num_envs=32, device=str(device), steps=steps, seed=71, small_terrain=True)
    trace = ST.load_trace(path, mmap=True)
    frames = torch.from_numpy(np.array(trace.frames))              # copy out of the read-only map
    env2.physics.live.copy_(frames[0])
    env2.physics.set_frames(frames[1:])
    for t in range(steps):
        rec2.feed(t)
        obs, rew, term, trunc, _ = env2.step(trace.actions[t].to(device))
        for a, b in zip((to_np(obs["policy"]), to_np(rew), to_np(term), to_np(trunc)), outs[t]):
            np.testing.assert_array_equal(a, b)
