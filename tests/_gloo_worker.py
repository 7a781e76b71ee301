"""world_size-2 gloo worker for tests/test_host_cpu.py::test_pcm_gather_over_gloo_world2."""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "qwen3-tts-rust_amd"))
from q3tts import dist as qd  # noqa: E402


def fake_pcm(gi):  # length and content are functions of the GLOBAL utterance index only
    return (np.arange(100 + 37 * gi, dtype=np.float32) * 0.001 + gi).astype(np.float32)


dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_total = 7
mine = [fake_pcm(gi) for gi in qd.shard_indices(n_total, rank, world)]
g = qd.gather_pcm(dist, mine, rank, world)
# the device-gather form bench.py uses on N > 1 GPUs (here on CPU tensors over gloo): packed rows + lengths, i16 on the wire
import torch  # noqa: E402
stride = 100 + 37 * n_total
rows = torch.zeros((len(mine), stride), dtype=torch.float32)
for i, a in enumerate(mine):
    rows[i, :a.size] = torch.from_numpy(a * 0.01)
gd = qd.gather_pcm_device(dist, rows, [a.size for a in mine], rank, world, as_i16=True)
if rank == 0:
    tens, lens = gd
    for r in range(world):
        idx = qd.shard_indices(n_total, r, world)
        assert int(lens[r, 0]) == len(idx)
        for j, gi in enumerate(idx):
            # the reference's conversion (src/utils/audio.rs:35-37), as tests/test_host_cpu.py states it for save_wav
            want = np.trunc(np.clip((fake_pcm(gi) * 0.01).astype(np.float32) * np.float32(32767.0), -32768.0, 32767.0)).astype(np.int16)
            assert int(lens[r, 1 + j]) == want.size and np.array_equal(tens[r][j, :want.size].numpy(), want), (r, j)
else:
    assert gd is None
if rank == 0:
    full = qd.reassemble(g, n_total, world)
    assert all(np.array_equal(full[i], fake_pcm(i)) for i in range(n_total))
    with open(os.environ["Q3_GLOO_OUT"], "w") as f:
        f.write(f"ok {n_total}")
else:
    assert g is None
dist.barrier()
dist.destroy_process_group()
