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


class _DevArray64:
    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i8", "data": (ptr, False), "version": 2}


def local_cts_tensor(run):
    """Zero-copy torch view (int64, cuda) of a QueryRun's locally folded ciphertexts [plane][2][N]."""
    import torch
    return torch.as_tensor(_DevArray64(run.local_cts_ptr(), run.local_cts_words()), device="cuda")


def reduce_scatter_partials(partial, rank, world):
    """Column-interleaved partial buffer (G = world contiguous chunks) -> this rank's summed chunk.
    nccl (= RCCL): one reduce_scatter; gloo (CPU tests) has no reduce_scatter: all_reduce + slice."""
    import torch
    import torch.distributed as dist
    chunk = partial.numel() // world
    if dist.get_backend() == "nccl":
        mine = torch.empty(chunk, dtype=partial.dtype, device=partial.device)
        dist.reduce_scatter_tensor(mine, partial, op=dist.ReduceOp.SUM)
        return mine
    dist.all_reduce(partial, op=dist.ReduceOp.SUM)
    return partial[rank * chunk:(rank + 1) * chunk].clone()


def gather_local(local, rank, world, dst=0):
    """The G locally folded results -> one [g][...] tensor on `dst` (None elsewhere)."""
    import torch
    import torch.distributed as dist
    out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous()) if dist.get_backend() == "nccl" else \
        dist.all_gather(list(out.view(world, -1).unbind(0)), local.contiguous())
    return out if rank == dst else None


def scatter_fold_query(run, db, rank, world, overlap=True, fold_per_plane=False):
    """The N > 1 answer path after sp_query_begin, rank-local view (north star): row-sharded
    sweep -> RCCL reduce-scatter of the partial Regev ciphertexts over the column axis -> every rank folds its
    columns with the top nu_2 - log2(world) selector bits -> all-gather of one ciphertext per plane per rank ->
    rank 0 folds the last log2(world) levels, packs and encodes.  Returns the response bytes on rank 0, None
    elsewhere.

    overlap=True: everything is ordered on the query's HIP streams (no host synchronisation until the response
    is copied out); the database is swept one plane per launch and the reduce-scatter of plane p (RCCL's own
    stream) runs while the later planes are swept; then one local fold over all planes.
    fold_per_plane=True additionally folds plane p on the second stream as soon as its exchange is done.  Measured
    with the exchange replaced by a local copy (scripts/archive/r02/rank_emulation.py flow, C2): the per-plane fold chains are
    latency-bound and serialise, 3.22 vs 2.64 ms per query at 8 shards, 4.51 vs 4.11 at 4, 7.08 vs 6.97 at 2 —
    hence off by default.  overlap=False: one sweep launch, one reduce-scatter, host
    synchronisation between the steps (the reference implementation of the same data flow)."""
    import torch
    import torch.distributed as dist
    planes = run.params.get("instances") * run.params.get("n") ** 2
    if not overlap:
        run.sweep_scatter(db, world)
        run.sync()
        mine = reduce_scatter_partials(partial_tensor(run), rank, world)
        torch.cuda.synchronize()
        run.fold_local(mine.data_ptr(), world)
        run.sync()
        gathered = gather_local(local_cts_tensor(run), rank, world, dst=0)
        torch.cuda.synchronize()
        return run.finish_gathered(gathered.data_ptr(), world) if rank == 0 else None
    part = partial_tensor(run)
    pw = part.numel() // planes
    chunk = pw // world
    nccl = dist.get_backend() == "nccl"
    main, second = torch.cuda.ExternalStream(run.stream()), torch.cuda.ExternalStream(run.stream2())
    with torch.cuda.stream(main):
        mine = torch.empty(planes * chunk, dtype=part.dtype, device=part.device)
    works = []
    for pl in range(planes):
        src, dst = part[pl * pw:(pl + 1) * pw], mine[pl * chunk:(pl + 1) * chunk]
        with torch.cuda.stream(main):
            run.sweep_scatter_plane(db, world, pl)
            if nccl:   # RCCL's stream waits for the sweep of this plane, not for the later ones
                work = dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, async_op=True)
            else:
                dist.all_reduce(src, op=dist.ReduceOp.SUM)
                dst.copy_(src[rank * chunk:(rank + 1) * chunk])
                work = None
        if fold_per_plane:
            with torch.cuda.stream(second):
                if work is not None:
                    work.wait()        # stream-level: the second stream waits for this plane's exchange
                else:
                    second.wait_stream(main)
            run.fold_local_plane(dst.data_ptr(), world, pl)   # second stream, beside the later planes' sweeps
        elif work is not None:
            works.append(work)
    if fold_per_plane:
        run.fold_local_join()
    else:
        with torch.cuda.stream(main):
            for w in works:
                w.wait()               # stream-level: the query stream waits for the exchanges, the host does not
        run.fold_local(mine.data_ptr(), world)
    with torch.cuda.stream(main):
        local = local_cts_tensor(run)
        gathered = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        if nccl:
            dist.all_gather_into_tensor(gathered, local)
        else:
            dist.all_gather(list(gathered.view(world, -1).unbind(0)), local.contiguous())
        out = run.finish_gathered(gathered.data_ptr(), world) if rank == 0 else None
    run.sync()  # mine / gathered are released by the caller's scope: nothing may still be reading them
    return out


def scatter_layout_index(num_per, planes, G, plane, r, crt, z, ii, N=2048):
    """flat index of output (plane, r, crt, z, ii) in the column-interleaved partial buffer
    (sweep.hip sweep_out_index): chunk ii % G, then [plane][r][crt][z][ii // G]"""
    npl = num_per // G
    chunk_words = planes * 4 * N * npl
    return (ii % G) * chunk_words + (((plane * 2 + r) * 2 + crt) * N + z) * npl + ii // G


def scatter_plane_layout_index(num_per, G, plane, r, crt, z, ii, N=2048):
    """flat index of output (plane, r, crt, z, ii) when the sweep runs one plane per launch
    (sp_query_sweep_scatter_plane): [plane][chunk ii % G][r][crt][z][ii // G]"""
    npl = num_per // G
    return plane * 4 * N * num_per + (ii % G) * 4 * N * npl + ((r * 2 + crt) * N + z) * npl + ii // G


def fold_schedule(nu_2, G):
    """Which GSW selector bits each phase consumes: local phase folds nu_2 - log2(G) levels with
    v_folding[nu_2-1 .. log2 G]; the final phase folds log2(G) levels with v_folding[log2 G - 1 .. 0]."""
    lg = G.bit_length() - 1
    assert 1 << lg == G and lg <= nu_2
    return list(range(nu_2 - 1, lg - 1, -1)), list(range(lg - 1, -1, -1))


def partial_layout_index(num_per, plane, r, crt, z, ii, N=2048):
    return (((plane * 2 + r) * 2 + crt) * N + z) * num_per + ii


def modulus_of_partial_index(idx, num_per, N=2048):
    """q_crt for flat indices into the partial buffer"""
    crt = (np.asarray(idx) // (N * num_per)) % 2
    return np.where(crt == 0, Q0, Q1)


# ------------------------------------------------------------------------------------------------
# The same flow with the exchange issued by the library itself (include/spiral_hip.h sp_comm_*,
# sdk_amd/csrc/comm.cpp: RCCL linked directly) -- what a non-Python host (lib/server, Rust) calls.
class Comm:
    """sp_comm_t: one per process / GPU.  Comm.rccl(rank, world, id128) joins the RCCL communicator whose id rank 0
    made with Comm.unique_id() (collective); Comm.custom(rank, world, reduce_scatter, all_gather) plugs host-supplied
    collectives in (callables (send_ptr, recv_ptr, count, hip_stream) -> 0)."""

    def __init__(self, h, keep=None):
        self.h, self._keep = h, keep

    @staticmethod
    def unique_id():
        import ctypes as C
        from .spiral import _chk, lib
        buf = (C.c_uint8 * 128)()
        _chk(lib().sp_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def rccl(cls, rank, world, id128):
        import ctypes as C
        from .spiral import SpiralError, _err, lib
        L = lib()
        L.sp_comm_create.restype = C.c_void_p
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(id128))
        h = L.sp_comm_create(C.c_int(rank), C.c_int(world), buf)
        if not h:
            raise SpiralError(_err())
        return cls(h)

    @classmethod
    def custom(cls, rank, world, reduce_scatter, all_gather):
        import ctypes as C
        from .spiral import SpiralError, _err, lib
        FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

        class Ops(C.Structure):
            _fields_ = [("reduce_scatter_u32", FN), ("all_gather_u64", FN), ("user", C.c_void_p)]

        def wrap(f):
            def g(user, send, recv, count, stream):
                try:
                    return int(f(send, recv, count, stream) or 0)
                except Exception as e:  # an exception must not unwind through the C frames
                    import sys
                    print("custom collective raised %r" % (e,), file=sys.stderr, flush=True)
                    return 1
            return FN(g)
        rs, ag = wrap(reduce_scatter), wrap(all_gather)
        ops = Ops(rs, ag, None)
        L = lib()
        L.sp_comm_create_custom.restype = C.c_void_p
        h = L.sp_comm_create_custom(C.c_int(rank), C.c_int(world), C.byref(ops))
        if not h:
            raise SpiralError(_err())
        return cls(h, keep=(rs, ag, ops))

    def free(self):
        import ctypes as C
        from .spiral import lib
        if getattr(self, "h", None):
            lib().sp_comm_free(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            from .spiral import _finalizing
            if not _finalizing():
                self.free()
        except Exception:
            pass

    def barrier(self):
        import ctypes as C
        from .spiral import _chk, lib
        _chk(lib().sp_comm_barrier(C.c_void_p(self.h)))

    def timings(self):
        import ctypes as C
        from .spiral import _chk, lib
        t = (C.c_float * 3)()
        _chk(lib().sp_comm_timings(C.c_void_p(self.h), t))
        return list(t)

    def describe(self):
        """sp_comm_describe: transport, RCCL version, collective sizes and the last query's timings, as a dict"""
        import ctypes as C
        import json
        from .spiral import _chk, lib
        buf = C.create_string_buffer(2048)
        _chk(lib().sp_comm_describe(C.c_void_p(self.h), buf, C.c_size_t(len(buf))))
        return json.loads(buf.value.decode())

    def process_query(self, params, pp, query, shard):
        """sp_process_query_sharded: the response bytes on rank 0, b"" elsewhere"""
        import ctypes as C
        from .spiral import _bytes, _chk, _p, lib, u8p
        d = _bytes(query.data if hasattr(query, "data") else bytes(query))
        n = params.get("response_bytes")
        out = np.zeros(n, dtype=np.uint8)
        ln = C.c_size_t(0)
        _chk(lib().sp_process_query_sharded(C.c_void_p(self.h), C.c_void_p(params.h), C.c_void_p(pp.h), _p(d, u8p),
                                            C.c_size_t(d.size), C.c_void_p(shard.h), _p(out, u8p), C.c_size_t(n),
                                            C.byref(ln)))
        return out[:ln.value].tobytes()


    def reserve(self, params):
        """sp_comm_reserve: exchange buffers for `params` allocated now (no allocation inside a sharded query)"""
        import ctypes as C
        from .spiral import _chk, lib
        _chk(lib().sp_comm_reserve(C.c_void_p(self.h), C.c_void_p(params.h)))

    def process_queries(self, params, pps, queries, shard):
        """sp_process_queries_sharded: a list of queries, query k + 1 expanding while query k is swept; the list of
        responses on rank 0, [] elsewhere"""
        import ctypes as C
        from .spiral import _bytes, _chk, _p, lib, u8p
        n = len(queries)
        if not isinstance(pps, (list, tuple)):
            pps = [pps] * n
        bufs = [_bytes(q.data if hasattr(q, "data") else bytes(q)) for q in queries]
        rb = params.get("response_bytes")
        out = np.zeros(max(1, n * rb), dtype=np.uint8)
        ln = C.c_size_t(0)
        pp_arr = (C.c_void_p * n)(*[pp.h for pp in pps])
        q_arr = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        l_arr = (C.c_size_t * n)(*[b.size for b in bufs])
        _chk(lib().sp_process_queries_sharded(C.c_void_p(self.h), C.c_void_p(params.h), pp_arr, q_arr, l_arr, C.c_int(n),
                                              C.c_void_p(shard.h), _p(out, u8p), C.c_size_t(rb), C.byref(ln)))
        if ln.value == 0:
            return []
        return [out[k * rb:(k + 1) * rb].tobytes() for k in range(n)]


class NullTransport:
    """Collectives that return at once (no data moves): the custom transport for timing ONE rank's critical path of a
    G-rank sharded query on a single GPU (scripts/archive/r03/r03_rank_critical_path.py).  Results are meaningless."""

    def __init__(self, rank, world):
        self.comm = Comm.custom(rank, world, lambda send, recv, count, stream: 0, lambda send, recv, count, stream: 0)


class LoopbackWorld:
    """G ranks as G host threads sharing ONE GPU, with in-process stand-ins for the two collectives (device sums /
    copies ordered with HIP events, ncclReduceScatter / ncclAllGather semantics).  Lets the whole N > 1 flow of
    sp_process_query_sharded -- G workspaces, 2 G streams, per-plane exchanges overlapping the next plane's sweep --
    run and be byte-checked on a 1-GPU box.  rank r's thread: world.comm(r).process_query(...)."""

    def __init__(self, world):
        import threading
        self.world = world
        self.bar = threading.Barrier(world, timeout=300)
        self.slot = [None] * world
        self.done = [None] * world
        self.comms = [Comm.custom(r, world, self._rs(r), self._ag(r)) for r in range(world)]

    def comm(self, r):
        return self.comms[r]

    def _exchange(self, r, send, stream, body):
        """publish (send, ready-event) -> rendezvous -> body(all sends) on my stream after everybody's data is ready ->
        rendezvous -> nobody's stream runs on before every reader of its buffer is done"""
        import torch
        st = torch.cuda.ExternalStream(stream)
        ev = torch.cuda.Event()
        ev.record(st)
        self.slot[r] = (send, ev)
        self.bar.wait()
        for (_, e) in self.slot:
            st.wait_event(e)
        with torch.cuda.stream(st):
            body([s for (s, _) in self.slot])
        fin = torch.cuda.Event()
        fin.record(st)
        self.done[r] = fin
        self.bar.wait()
        for e in self.done:
            st.wait_event(e)
        self.bar.wait()   # slots may be overwritten by the next collective only after everybody has read them
        return 0

    def _rs(self, r):
        def f(send, recv, count, stream):
            import torch

            def body(sends):
                out = torch.as_tensor(_DevArray(recv, count), device="cuda")
                acc = None
                for s in sends:
                    t = torch.as_tensor(_DevArray(s + 4 * count * r, count), device="cuda")
                    acc = t.clone() if acc is None else acc.add_(t)
                out.copy_(acc)
            return self._exchange(r, send, stream, body)
        return f

    def _ag(self, r):
        def f(send, recv, count, stream):
            import torch

            def body(sends):
                out = torch.as_tensor(_DevArray64(recv, count * self.world), device="cuda")
                for g, s in enumerate(sends):
                    out[g * count:(g + 1) * count].copy_(torch.as_tensor(_DevArray64(s, count), device="cuda"))
            return self._exchange(r, send, stream, body)
        return f

    def run(self, fn):
        """fn(rank) on every rank thread; returns the list of results (re-raises the first exception)"""
        import threading
        res, err = [None] * self.world, []

        def go(r):
            try:
                res[r] = fn(r)
            except Exception as e:  # noqa: BLE001
                err.append(e)
                self.bar.abort()
        ts = [threading.Thread(target=go, args=(r,)) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if err:
            raise err[0]
        return res
