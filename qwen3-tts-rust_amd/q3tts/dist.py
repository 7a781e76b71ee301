"""Multi-GPU plumbing: utterance sharding (no data-path collective) and the one PCM gather.

The hot path shards over independent utterances: rank r of G owns the global indices
{i : i mod G == r}; every utterance's sampler stream is a function of its GLOBAL index only, so results do not
depend on G. The only collective is the gather of the variable-length PCM to rank 0 (RCCL over xGMI on the GPU
box, gloo in the CPU tests): an all_gather of the lengths followed by a padded gather.
"""
import numpy as np


def shard_indices(n_total, rank, world):
    """Global utterance indices owned by `rank` (round-robin, order preserving)."""
    return list(range(rank, n_total, world))


def global_seed(base_seed, global_index):
    """Per-utterance sampler seed: depends on the global index only (never on rank / world size)."""
    return int(base_seed) + int(global_index)


def workload_class(global_index, n_classes):
    """Length class of an utterance of the weak-scaling workload: c(i) = (i + i // n) mod n, n = 64 (bench.py: a constant, the
    benchmark's utterances per GPU, so that an utterance never depends on the rank count or on --batch).

    An utterance's prompt and forced length are functions of its CLASS, its sampler seed of its global index. Rank r of G owns
    {r + G j : j < n}; when G divides n those indices hit every class exactly once (i = r + n a + G b with j = (n / G) a + b gives
    c = (r + G b + a) mod n, a bijection onto [0, n) for b < n / G, a < G), so every rank — and a single GPU — runs the same n lengths
    with different sampler streams and a 1 -> G curve measures the hardware, not the draw (with lengths drawn
    per global index the 512-utterance list held 8 715 frames per rank against 9 865 for the first 64, a 0.883 'efficiency' before
    any hardware effect). Counterpart: the reference runs one utterance at a time (src/models/llama/mod.rs:413, n_seq_max = 1)."""
    i, n = int(global_index), int(n_classes)
    return (i + i // n) % n


def gather_pcm(dist, pcm_list, rank, world, device="cpu", dtype=None, to_numpy=True):
    """Gathers per-utterance PCM arrays of every rank to rank 0.

    pcm_list: list of 1-D float32 numpy arrays (this rank's utterances, in shard order).
    Returns on rank 0: list (over ranks) of lists of numpy arrays; on other ranks: None.
    to_numpy=False (bench.py): rank 0 keeps what the collective delivered — (gathered tensors, lengths, counts) on `device` —
    instead of copying 8 x 64 utterances back to the host one by one inside the timed region.
    """
    import torch
    dtype = dtype or torch.float32
    n_local = len(pcm_list)
    lens = torch.tensor([a.size for a in pcm_list], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n_local], dtype=torch.int64, device=device))
    max_n = int(max(int(c.item()) for c in counts))
    lens_pad = torch.zeros(max_n, dtype=torch.int64, device=device)
    lens_pad[:n_local] = lens
    all_lens = [torch.zeros(max_n, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_lens, lens_pad)
    max_len = max(1, int(torch.stack(all_lens).max().item()))
    buf = torch.zeros((max_n, max_len), dtype=dtype, device=device)
    for i, a in enumerate(pcm_list):
        if a.size:
            buf[i, :a.size] = torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    if rank != 0:
        return None
    if not to_numpy:
        return gathered, all_lens, counts
    out = []
    for r in range(world):
        n_r = int(counts[r].item())
        out.append([gathered[r][i, :int(all_lens[r][i].item())].float().cpu().numpy() for i in range(n_r)])
    return out


class _DevArray:
    """A device allocation of libq3tts seen through __cuda_array_interface__ (zero-copy torch view)."""
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def device_pcm_tensor(engine, device):
    """torch view [rows][stride] f32 of the engine's packed device PCM of the last batch (q3tts_get_device_pcm)."""
    import torch
    ptr, stride, n = engine.device_pcm()
    if not ptr or n <= 0:
        return torch.zeros((0, 1), dtype=torch.float32, device=device)
    return torch.as_tensor(_DevArray(ptr, (n, stride)), device=device)


def gather_pcm_device(dist, pcm_rows, n_samples, rank, world, as_i16=True):
    """The path's ONE collective, straight from device memory: pcm_rows [n][stride] f32 on the device (row i =
    utterance i of this rank), n_samples[i] valid samples each. Lengths travel in one all_gather, the PCM (i16 by the reference's own
    conversion, src/utils/audio.rs:35-37: scale, clamp to [-32768, 32767], truncate; or f32) in one padded gather to rank 0. Returns on rank 0 (list of per-rank tensors
    [max_n][max_len], lengths [world][max_n]); elsewhere None. Works over gloo as well (CPU tensors)."""
    import torch
    dev = pcm_rows.device
    n_local = len(n_samples)
    lens = torch.tensor([n_local] + [int(x) for x in n_samples], dtype=torch.int64, device=dev)
    cnt = torch.tensor([n_local], dtype=torch.int64, device=dev)
    cmax = cnt.clone()
    dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
    max_n = int(cmax.item())
    lens_pad = torch.zeros(max_n + 1, dtype=torch.int64, device=dev)
    lens_pad[:n_local + 1] = lens
    all_lens = [torch.zeros_like(lens_pad) for _ in range(world)]
    dist.all_gather(all_lens, lens_pad)
    max_len = max(1, int(torch.stack(all_lens)[:, 1:].max().item()))
    buf = torch.zeros((max_n, max_len), dtype=torch.int16 if as_i16 else torch.float32, device=dev)
    if n_local:
        blk = pcm_rows[:n_local, :max_len]
        mask = torch.arange(max_len, device=dev)[None, :] < lens[1:, None]
        blk = torch.where(mask, blk, torch.zeros((), dtype=blk.dtype, device=dev))
        # f32 -> i16 exactly as the reference saves audio (src/utils/audio.rs:35-37): (x * 32767).clamp(-32768, 32767) as i16,
        # i.e. truncation toward zero — the same expression as q3tts/api.py's save_wav
        buf[:n_local] = torch.trunc((blk * 32767.0).clamp(-32768.0, 32767.0)).to(torch.int16) if as_i16 else blk
    wire = buf.view(torch.uint8)  # neither RCCL nor gloo has a 16-bit integer type: the i16 samples travel as bytes
    gathered = [torch.zeros_like(wire) for _ in range(world)] if rank == 0 else None
    dist.gather(wire, gathered, dst=0)
    if rank != 0:
        return None
    return [g.view(buf.dtype) for g in gathered], torch.stack(all_lens)


def reassemble(gathered, n_total, world):
    """Inverse of shard_indices on rank 0: list indexed by global utterance index."""
    res = [None] * n_total
    for r in range(world):
        for j, gi in enumerate(shard_indices(n_total, r, world)):
            res[gi] = gathered[r][j]
    return res
