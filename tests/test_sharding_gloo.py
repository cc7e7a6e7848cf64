"""N > 1 path on CPU (gloo, world_size 2 and 4): the row-shard split + the one exchange step.
Each rank computes its shard's partial first-dimension residues with the oracle, the partial buffers are
summed onto rank 0 by sdk_amd.sharding.reduce_partials (the call bench.py makes over RCCL), and the result
must equal the unsharded multiply_reg_by_database."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 2048
Q0, Q1 = 268369921, 249561089


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dim0, num_per, nz, out_path):
    sys.path.insert(0, ROOT)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch
    import torch.distributed as dist
    import oracle
    from sdk_amd.sharding import modulus_of_partial_index, partial_layout_index, reduce_partials, shard_rows
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.default_rng(7)  # same inputs on every rank
    db = rng.integers(0, Q0, nz * num_per * dim0, dtype=np.uint64) | (rng.integers(0, Q1, nz * num_per * dim0, dtype=np.uint64) << np.uint64(32))
    qv = rng.integers(0, Q0, nz * dim0 * 2, dtype=np.uint64) | (rng.integers(0, Q1, nz * dim0 * 2, dtype=np.uint64) << np.uint64(32))
    j0, j1 = shard_rows(dim0, rank, world)
    nj = j1 - j0
    db_s = np.ascontiguousarray(db.reshape(nz, num_per, dim0)[:, :, j0:j1])
    qv_s = np.ascontiguousarray(qv.reshape(nz, dim0, 2)[:, j0:j1, :])
    part = oracle.sweep_rows(db_s, qv_s, nz, nj, num_per)  # [z][ii][n0_0, n0_1, n1_0, n1_1]
    # scatter into the library's partial layout [plane][r][crt][z][ii] (plane 0, z < nz)
    buf = np.zeros(4 * N * num_per, dtype=np.int32)
    zz, ii = np.meshgrid(np.arange(nz), np.arange(num_per), indexing="ij")
    for which, (r, crt) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)]):
        buf[partial_layout_index(num_per, 0, r, crt, zz, ii)] = part[:, :, which].astype(np.int32)
    t = torch.from_numpy(buf)
    reduce_partials(t, dst=0)
    if rank == 0:
        total = t.numpy().astype(np.int64)
        idx = np.arange(total.size)
        red = (total % modulus_of_partial_index(idx, num_per)).astype(np.uint64)
        full = oracle.sweep_rows(db, qv, nz, dim0, num_per)
        ok = True
        for which, (r, crt) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)]):
            ok &= bool((red[partial_layout_index(num_per, 0, r, crt, zz, ii)] == full[:, :, which]).all())
        ok &= int(total.max()) < 2**31
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dim0,num_per", [(2, 64, 4), (4, 512, 8), (2, 512, 128)])
def test_row_shard_reduce_gloo(tmp_path, world, dim0, num_per):
    import torch.multiprocessing as mp
    out = tmp_path / "res.txt"
    mp.spawn(_worker, args=(world, _free_port(), dim0, num_per, 6, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


def _worker_dist_fold(rank, world, port, out_path):
    """The distributed-fold algorithm of bench.py (N > 1) with the oracle standing in for the kernels."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch
    import torch.distributed as dist
    import oracle
    from conftest import FAST
    from sdk_amd.sharding import (fold_schedule, gather_local, reduce_scatter_partials, scatter_layout_index,
                                  shard_rows)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = dict(FAST, nu_2=3)
    o = oracle.Params(cfg)
    cl = oracle.Client(o)
    pp = cl.generate_keys(5)
    idx = 333
    q = cl.generate_query(idx, 6)
    item, db = o.generate_random_db_and_get_item(idx)
    v_reg, v_fold = o.expand_query(pp, q)
    v_neg = o.get_v_folding_neg(v_fold)
    dim0, num_per, planes, nu2 = o.dim0, o.num_per, 4, o.db_dim_2
    j0, j1 = shard_rows(dim0, rank, world)
    nj = j1 - j0
    npl = num_per // world
    buf = np.zeros(planes * 4 * N * num_per, dtype=np.int32)
    zz, ii = np.meshgrid(np.arange(N), np.arange(num_per), indexing="ij")
    dbp = db.reshape(planes, N, num_per, dim0)
    qv = v_reg.reshape(N, dim0, 2)
    for pl in range(planes):
        part = oracle.sweep_rows(np.ascontiguousarray(dbp[pl][:, :, j0:j1]), np.ascontiguousarray(qv[:, j0:j1, :]), N, nj, num_per)
        for which, (r, crt) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)]):
            buf[scatter_layout_index(num_per, planes, world, pl, r, crt, zz, ii)] = part[:, :, which].astype(np.int32)
    mine = reduce_scatter_partials(torch.from_numpy(buf), rank, world).numpy().astype(np.int64)
    mine = mine.reshape(planes, 2, 2, N, npl)
    mine[:, :, 0] %= Q0
    mine[:, :, 1] %= Q1
    local_idx, final_idx = fold_schedule(nu2, world)
    W = 2 * 2 * o.t_gsw * 2 * N  # words per GSW ct
    lg = world.bit_length() - 1
    local = []
    for pl in range(planes):
        # ct i of this rank = column g + G*i : NTT form [r][crt][z]
        cts_ntt = np.ascontiguousarray(mine[pl].transpose(3, 0, 1, 2)).astype(np.uint64).reshape(-1)
        raw = o.from_ntt(cts_ntt)
        k = len(local_idx)
        if k:
            raw = o.fold_ciphertexts(raw, v_fold[lg * W:nu2 * W], v_neg[lg * W:nu2 * W], nu=k)
        local.append(raw[:2 * N])
    gathered = gather_local(torch.from_numpy(np.concatenate(local).astype(np.int64)), rank, world, dst=0)
    if rank == 0:
        g = gathered.numpy().astype(np.uint64).reshape(world, planes, 2 * N)
        ok = True
        slice_words = dim0 * num_per * N
        for pl in range(planes):
            cts = np.ascontiguousarray(g[:, pl]).reshape(-1)
            res = o.fold_ciphertexts(cts, v_fold[:lg * W], v_neg[:lg * W], nu=lg)[:2 * N] if lg else cts[:2 * N]
            full = o.from_ntt(o.multiply_reg_by_database(db[pl * slice_words:(pl + 1) * slice_words], v_reg))
            exp = o.fold_ciphertexts(full, v_fold, v_neg)[:2 * N]
            ok &= bool((res == exp).all())
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_distributed_fold_gloo(tmp_path, world):
    """reduce-scatter over columns + local fold + gather + final fold == the single-process fold tree."""
    import torch.multiprocessing as mp
    out = tmp_path / "res.txt"
    mp.spawn(_worker_dist_fold, args=(world, _free_port(), str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


def test_shard_rows_contract():
    from sdk_amd.sharding import shard_rows
    assert shard_rows(512, 0, 8) == (0, 64) and shard_rows(512, 7, 8) == (448, 512)
    with pytest.raises(ValueError):
        shard_rows(512, 0, 16)      # int32 partial sums are only overflow-free for <= 8 shards
    with pytest.raises(ValueError):
        shard_rows(6, 0, 4)


@pytest.mark.parametrize("num_per,G", [(16, 2), (32, 8), (8, 4)])
def test_scatter_layouts_are_permutations(num_per, G):
    """Both column-interleaved partial layouts ([g][plane].. of sweep_scatter and [plane][g].. of
    sweep_scatter_plane) enumerate the buffer exactly once, and rank g's chunk of every plane holds the columns
    ii = g (mod G) in [r][crt][z][ii // G] order — what sp_query_fold_local consumes."""
    from sdk_amd.sharding import scatter_layout_index, scatter_plane_layout_index
    N, planes = 8, 4   # the index functions are linear in N: a small N keeps the enumeration cheap
    total = planes * 4 * N * num_per
    a = np.zeros(total, dtype=np.int64)
    b = np.zeros(total, dtype=np.int64)
    npl = num_per // G
    for pl in range(planes):
        for r in range(2):
            for c in range(2):
                for z in range(N):
                    for ii in range(num_per):
                        a[scatter_layout_index(num_per, planes, G, pl, r, c, z, ii, N=N)] += 1
                        i2 = scatter_plane_layout_index(num_per, G, pl, r, c, z, ii, N=N)
                        b[i2] += 1
                        # inside plane pl, chunk g starts at g * (plane words / G)
                        g, off = ii % G, i2 - pl * 4 * N * num_per
                        assert off // (4 * N * npl) == g
                        assert off % (4 * N * npl) == ((r * 2 + c) * N + z) * npl + ii // G
    assert (a == 1).all() and (b == 1).all()
