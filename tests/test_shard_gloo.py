"""N > 1 path on CPU: two processes over gloo exercise the frequency sharding
and the end-of-run gather used by bench.py --gpus N (there over RCCL)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    from emg3d_amd import shard, meshes, fields
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    freqs = {freqs!r}
    mine = shard.my_frequencies(freqs, rank, world)
    h = meshes.stretched_widths(4, 2, 50., 1.2)
    grid = meshes.TensorMesh([h, h, h], origin=(-h.sum() / 2,) * 3)
    # stand-in for the per-frequency solve: the source field itself
    local = [np.array(fields.get_source_field(grid, [0., 0., 0., 30., 10.], f)) for f in mine]
    allf = shard.gather_fields(local)
    assert len(allf) == world
    got = {{}}
    for r in range(world):
        for f, a in zip(shard.my_frequencies(freqs, r, world), allf[r]):
            got[f] = a
    assert sorted(got) == sorted(freqs)
    for f in freqs:
        ref = np.array(fields.get_source_field(grid, [0., 0., 0., 30., 10.], f))
        assert np.array_equal(got[f], ref), f
    # a sharded survey's responses: (n_src, my frequencies, n_rec) per rank -> (n_src, all frequencies, n_rec) everywhere
    ns, nr = 3, 4
    full = np.array([[[complex(100 * i + 10 * k + j, -f) for j in range(nr)] for k, f in enumerate(freqs)] for i in range(ns)]
                    ).reshape(ns, len(freqs), nr)
    local = full[:, rank::world, :]
    got = shard.gather_survey(local, freqs)
    assert got.shape == (ns, len(freqs), nr) and np.array_equal(got, full)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


@pytest.mark.parametrize("freqs", [[0.25, 0.5, 1.0], [1.0], []])
def test_two_rank_gloo_gather(tmp_path, freqs):
    """3 frequencies on 2 ranks (counts 2 / 1), ONE frequency on 2 ranks (rank 1 owns nothing and must still
    take part in the collectives) and no frequency at all."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, freqs=freqs))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_round_robin():
    from emg3d_amd import shard
    f = [1, 2, 3, 4, 5]
    assert shard.my_frequencies(f, 0, 2) == [1., 3., 5.]
    assert shard.my_frequencies(f, 1, 2) == [2., 4.]
    assert sum((shard.my_frequencies(f, r, 8) for r in range(8)), []) == [1., 2., 3., 4., 5.]
