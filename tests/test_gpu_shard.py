"""GPU: the multi-GPU harness on ONE GPU -- a 1-rank `nccl` (= RCCL) process group runs the device-resident
end-of-run gather (`shard.gather_efield_device`: all_gather_into_tensor straight out of the handle's HBM
buffer, ordered behind the handle's stream by a stream wait) and the host-staged `shard.gather_fields`; both must return
exactly `get_efield()`.  Runs in a child process (own process group, own HIP context)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import emg3d_amd as em
    from emg3d_amd import shard
    from emg3d_amd.solver import DeviceMG, MGParameters
    h = em.meshes.stretched_widths(8, 4, 100., 1.3)
    grid = em.TensorMesh([h, h, h], origin=(-h.sum() / 2,) * 3)
    rho = 10 ** np.random.default_rng(0).uniform(-0.5, 1.5, grid.nC)
    model = em.Model(grid, rho, 2 * rho, 3 * rho)
    for freq in (1.0, -2.0):                 # complex128 and float64 (Laplace) handles
        sfield = em.get_source_field(grid, [0., 0., 0., 30., 10.], freq)
        vm = em.VolumeModel(grid, model, sfield)
        var = MGParameters(verb=0, cycle='F', sslsolver=False, linerelaxation=True, semicoarsening=True,
                           vnC=grid.vnC)
        with DeviceMG(grid, vm, sfield.dtype) as dev:
            dev.set_params(var); dev.set_sfield(sfield); dev.set_efield(None)
            dev.cycles(3, [1, 2, 3], [4, 5, 6])          # no host sync between the cycles and the gather
            allf = shard.gather_efield_device(dev)
            torch.cuda.synchronize()
            e = dev.get_efield()
            assert np.abs(e).max() > 0
            got = allf[0].cpu().numpy()
            got = got.view(np.complex128) if sfield.dtype == np.complex128 else got
            assert allf.shape[0] == 1 and np.array_equal(got, e), freq
            # zero copy: the tensor IS the handle's buffer
            assert shard.efield_tensor(dev).data_ptr() == dev.efield_devptr
            host = shard.gather_fields(e)
            assert len(host) == 1 and np.array_equal(host[0][0], e)
    # the shard's payload shrunk to the receiver responses (SURVEY 8f rank 3): solve two frequencies without
    # ever downloading a field, gather 16 bytes per receiver
    rec = (np.array([-150., 40., 220.]), np.array([30., -80., 10.]), np.array([-60., -120., 90.]), 25., 10.)
    # ... "never": DeviceMG.get_efield (the nE-sized download) must not run at all on this path
    downloads = []
    orig_get = DeviceMG.get_efield
    DeviceMG.get_efield = lambda self, out=None: (downloads.append(1), orig_get(self, out))[1]
    res = shard.solve_frequencies(grid, model, [0., 0., 0., 30., 10.], [0.5, 2.0], rec=rec, return_field=False,
                                  cycle='F', semicoarsening=True, linerelaxation=True, verb=0)
    DeviceMG.get_efield = orig_get
    assert not downloads, "solve_frequencies(return_field=False) downloaded a field"
    assert all(r[0] is None and r[1]['exit'] == 0 for r in res)
    allr = shard.gather_fields([r[2] for r in res])
    assert len(allr) == 1 and len(allr[0]) == 2
    full = shard.solve_frequencies(grid, model, [0., 0., 0., 30., 10.], [0.5, 2.0], cycle='F', semicoarsening=True,
                                   linerelaxation=True, verb=0)
    for got, (e, _) in zip(allr[0], full):
        ref = em.get_receiver_response(grid, e, rec)
        assert np.allclose(got, ref, rtol=1e-13, atol=0), (got, ref)
    # a survey: sources x frequencies; this rank's frequencies, the sources of each batched through the same launches
    srcs = [[0., 0., 0., 30., 10.], [120., -60., 40., 75., -5.], [-200., 90., -30., 10., 0.]]
    mine = shard.my_frequencies([0.5, 2.0], dist.get_rank(), dist.get_world_size())
    resp, infos = shard.solve_survey(grid, model, srcs, mine, rec, batch=2, cycle='F', semicoarsening=True,
                                     linerelaxation=True, verb=0)
    assert resp.shape == (3, 2, 3) and all(i['exit'] == 0 for row in infos for i in row)
    # (both paths form eta = (smu0 * V) * sigma on the device, VolumeModel's rounding, and a batched system is bit for bit
    # its own solve: the survey's responses ARE the per-frequency ones)
    assert np.array_equal(resp[0, 0], res[0][2]) and np.array_equal(resp[0, 1], res[1][2])
    assert np.array_equal(shard.gather_survey(resp, [0.5, 2.0]), resp)      # one rank: the whole survey
    dist.barrier()
    dist.destroy_process_group()
    print("nccl gather ok")
""")


def test_one_rank_nccl_device_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "nccl gather ok" in p.stdout


def test_bench_distributed_path_one_rank(tmp_path):
    """bench.py's N > 1 code path on the one GPU of the test box: EMG3D_FORCE_DIST=1 -> 1-rank RCCL group,
    per-rank times through a collective, device-resident gather.  (The rank spawner of `--gpus N` itself is
    covered on CPU: tests/test_host_logic.py::test_bench_spawns_ranks.)"""
    import json
    env = dict(os.environ, EMG3D_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "32F", "--steps", "2",
                        "--warmup", "1", "--no-cpu", "--no-256", "--no-tol", "--multi", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["gather_ms"] > 0 and line["value"] > 0
    assert line["roofline"]["kernel"].startswith("k_line_sweep")
    # the record that shows which process group ran: RCCL's own world size, backend, devices
    assert line["rccl_world"] == 1 and line["dist_backend"] == "nccl" and line["device_count"] >= 1
    assert len(line["per_rank_device"]) == 1 and line["per_rank_device"][0]["rank"] == 0 and line["per_rank_device"][0]["name"]


def test_bench_two_ranks_end_to_end_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` end to end -- the rank spawner, two rank processes, barrier-bracketed timing, max over
    ranks through a collective, the device-resident gather of both fields -- on the ONE GPU of the test box:
    EMG3D_BENCH_SHARE_GPU=1 puts both ranks on GPU 0 and swaps RCCL (which refuses two ranks on one device) for gloo;
    everything else is the code path of the driver's N > 1 runs."""
    import json
    env = dict(os.environ, EMG3D_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "32F", "--steps", "2",
                        "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and len(line["per_rank_ms_per_step"]) == 2 and line["scaling"] == "weak"
    assert line["ms_per_step"] == pytest.approx(max(line["per_rank_ms_per_step"]))
    # whole-job aggregate: both ranks' cells over the slowest rank's time
    assert line["value"] == pytest.approx(2 * line["config"]["cells"] / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-6)
    assert line["gather_ms"] > 0 and line["gather_bytes_per_rank"] > 0 and 0 < line["value_incl_gather"] < line["value"]
    assert line["rccl_world"] == 2 and line["dist_backend"] == "gloo"
    assert [d["rank"] for d in line["per_rank_device"]] == [0, 1] and all(d["device"] == 0 for d in line["per_rank_device"])


def test_bench_eight_ranks_end_to_end_on_one_gpu(tmp_path):
    """`python bench.py --gpus 8` -- the shape of the driver's 8-GPU run (BASELINE configs[4]: eight frequencies, one per
    rank) -- with all eight rank processes on the ONE GPU of the test box (EMG3D_BENCH_SHARE_GPU=1, gloo instead of RCCL):
    eight fresh child processes, eight handles, eight different frequencies, the max over eight per-rank times, the gather
    with eight rows, one residual history per rank.  When real GPUs appear the same command runs unchanged."""
    import json
    env = dict(os.environ, EMG3D_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "32F", "--steps", "2",
                        "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and len(line["per_rank_ms_per_step"]) == 8 and line["scaling"] == "weak"
    assert len(set(line["per_rank_freq_Hz"])) == 8
    assert line["ms_per_step"] == pytest.approx(max(line["per_rank_ms_per_step"]))
    assert line["value"] == pytest.approx(8 * line["config"]["cells"] / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-6)
    assert line["gather_ms"] > 0 and line["gather_bytes_per_rank"] > 0
    # the job including its one collective: the same cells over (the slowest rank's cycles + the gather)
    assert line["value_incl_gather"] == pytest.approx(
        8 * line["config"]["cells"] * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"] + line["gather_ms"] * 1e-3) / 1e6, rel=1e-6)
    assert 0 < line["value_incl_gather"] < line["value"]
    hist = line["per_rank_rel_error_after"]
    assert len(hist) == 8 and all(len(h) == 3 and h[-1] < h[0] for h in hist)
    assert len({tuple(h) for h in hist}) == 8          # eight different systems
    assert "config_256V" not in line and "batched_sources" not in line      # single-GPU extras stay out of N > 1 lines


def test_bench_two_ranks_histories_equal_single_rank_runs(tmp_path):
    """The N-rank line against N single-rank runs (VERDICT r2, item 6 iii): `bench.py --gpus 2 --workload 64F` (both
    ranks on the one GPU of the test box, EMG3D_BENCH_SHARE_GPU=1) reports every rank's residual history; rank r's
    must equal, bit for bit, the history of a single-rank run of ITS frequency (`--freq-offset r`): the ranks are
    independent systems, sharing a GPU or a node must not change a digit.  Also: one host thread per rank for the
    BLAS / OpenMP pools."""
    import json
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "64F", "--steps", "4", "--warmup", "2",
            "--no-cpu", "--no-256", "--no-tol", "--batch", "0"]
    env = dict(os.environ, EMG3D_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "OMP_NUM_THREADS"):
        env.pop(k, None)
    p = subprocess.run(base + ["--gpus", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    two = json.loads(p.stdout.strip().splitlines()[-1])
    assert two["n_gpus"] == 2 and len(two["per_rank_rel_error_after"]) == 2
    assert two["host_threads_per_rank"] == "1"
    assert two["per_rank_freq_Hz"][0] != two["per_rank_freq_Hz"][1]
    assert len(two["per_rank_cycles_to_tol"]) == 2 and "roofline" in two and two["gather_ms"] > 0
    env1 = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "EMG3D_BENCH_SHARE_GPU"):
        env1.pop(k, None)
    for r in range(2):
        q = subprocess.run(base + ["--freq-offset", str(r)], env=env1, capture_output=True, text=True, timeout=900)
        assert q.returncode == 0, q.stdout[-3000:] + q.stderr[-3000:]
        one = json.loads(q.stdout.strip().splitlines()[-1])
        assert one["per_rank_freq_Hz"] == [two["per_rank_freq_Hz"][r]]
        assert one["per_rank_rel_error_after"][0] == two["per_rank_rel_error_after"][r], r
    # the two frequencies are different systems: their histories differ
    assert two["per_rank_rel_error_after"][0] != two["per_rank_rel_error_after"][1]


def test_concurrent_handles_with_the_placement_search_are_bitwise():
    """Three frequencies at 200^3 -- working copies of 385 MB: every handle runs the placement search -- on three host
    threads at once (`solve_frequencies(concurrent=3)`: three handles timing candidate blocks beside each other, one block pool)
    against one at a time (one handle re-targeted per frequency): same cycle counts and norms, bit-identical fields."""
    import numpy as np
    import bench
    import emg3d_amd as em
    from emg3d_amd import shard, _lib
    grid, model, sfield, cycle = bench.build_problem(em, "200V", 1.0)
    freqs = [1.0, 0.5, 2.0]
    out = {}
    for conc in (3, 1):
        _lib.load().emg3d_hip_release_cached()
        res = shard.solve_frequencies(grid, model, [0., 0., 0., 30., 10.], freqs, concurrent=conc, cycle=cycle, semicoarsening=True,
                                      linerelaxation=True, maxit=3, verb=0)
        out[conc] = [(np.array(e), info['it_mg'], float(info['abs_error'])) for e, info in res]
    for (e3, i3, a3), (e1, i1, a1) in zip(out[3], out[1]):
        assert i3 == i1 == 3 and a3 == a1 and np.isfinite(a3)
        assert np.array_equal(e3, e1)
