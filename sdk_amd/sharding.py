"""Row sharding of the first dimension across GPUs (north star): shard s of G holds
rows j in [s*dim0/G, (s+1)*dim0/G) of every (plane, z, ii); each shard's sweep yields partial residues
< q; their element-wise sum over shards (< G*q <= 2^31 for G <= 8) reduced mod q is the unsharded
first-dimension output.  The only exchange step is that one sum (RCCL ncclSum on int32 over xGMI)."""
import numpy as np

Q0, Q1 = 268369921, 249561089
MAX_SHARDS = 8  # 8 * (q0 - 1) < 2^31: int32 partial sums cannot overflow


def shard_rows(dim0, shard, num_shards):
    """[j0, j1) of first-dimension rows held by `shard` (include/spiral_hip.h sp_db_create)."""
    if not (1 <= num_shards <= MAX_SHARDS) or dim0 % num_shards:
        raise ValueError("dim0 must be divisible by num_shards <= %d" % MAX_SHARDS)
    nj = dim0 // num_shards
    return shard * nj, (shard + 1) * nj


class _DevArray:
    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def partial_tensor(run):
    """Zero-copy torch view (int32, cuda) of a QueryRun's partial first-dimension buffer,
    layout [plane][r][crt][z][ii]."""
    import torch
    return torch.as_tensor(_DevArray(run.partial_ptr(), run.partial_words()), device="cuda")


def reduce_partials(partial, dst=0):
    """Sum the per-shard partial buffers onto rank `dst` (torch.distributed: nccl = RCCL on GPUs, gloo in
    the CPU tests).  In place on `dst`."""
    import torch.distributed as dist
    dist.reduce(partial, dst=dst, op=dist.ReduceOp.SUM)
    return partial


def partial_layout_index(num_per, plane, r, crt, z, ii, N=2048):
    return (((plane * 2 + r) * 2 + crt) * N + z) * num_per + ii


def modulus_of_partial_index(idx, num_per, N=2048):
    """q_crt for flat indices into the partial buffer"""
    crt = (np.asarray(idx) // (N * num_per)) % 2
    return np.where(crt == 0, Q0, Q1)
