"""sdk_amd -- MI355X-native Spiral PIR server answer path (drop-in for lib/spiral-rs's server half).

The compute lives in ``libspiral_hip.so`` (hand-written gfx950 HIP kernels behind the C ABI declared in
``include/spiral_hip.h``); this package is the thin host-side mirror of the spiral-rs API surface
(``Params`` / ``PolyMatrixRaw`` / ``PolyMatrixNTT`` / ``PublicParameters`` / ``Query`` / ``server.*``).
There is no CPU fallback: every compute entry point raises if the library or a GPU is missing.
"""
from .spiral import (
    add,
    add_into,
    scalar_multiply,
    bench_sweep_batch,  # noqa: F401
    Database,
    Params,
    PolyMatrixNTT,
    PolyMatrixRaw,
    PublicParameters,
    Query,
    QueryRun,
    Server,
    NotFound,
    SpiralError,
    build_library,
    coefficient_expansion,
    encode,
    expand_query,
    fold_ciphertexts,
    fold_ciphertexts_fused,
    from_ntt,
    get_v_folding_neg,
    lib,
    library_path,
    multiply,
    multiply_reg_by_database,
    ntt_forward,
    ntt_inverse,
    pack,
    params_from_json,
    paths_taken,
    process_query,
    process_query_batch,
    regev_to_gsw,
    reorient_reg_ciphertexts,
    to_ntt,
)
