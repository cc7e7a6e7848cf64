// Phase program: a chain of small dependent launches of the answer path as ONE launch of a persistent grid with
// device-wide barriers between the phases (kernels.hpp has the rationale and the API).
//
// Reference structure served: the serial spine of coefficient_expansion (server.rs:19-121: round r needs round r - 1, and
// inside a round automorph -> gadget digits + NTT -> key-switch multiply-accumulate are dependent), regev_to_gsw
// (server.rs:123-151), get_v_folding_neg (server.rs:505-523), the small levels of fold_ciphertexts (server.rs:388-427),
// pack (server.rs:429-468) and encode (server.rs:470-503).
//
// Device-wide barrier: one monotonic 32-bit counter per Program; a workgroup arrives with an agent-scope release
// (write-back of its XCD's L2: the eight L2s of the part are not coherent with each other for ordinary lines) + atomic
// add, spins with s_sleep until the counter has reached generation * grid (compared as a signed difference, so the
// counter may wrap), then an agent-scope acquire (L1 / L2 invalidate).  The grid is at most program_wgs (default 2)
// workgroups per CU with <= 128 VGPRs and 18 KiB of LDS each -- co-resident four times over on an idle device -- and the
// launches of ALL programs of the process are chained by events (program_launch), so that two programs never wait for each
// other's slots.  A workgroup that spins for more than ~2 s traps instead of hanging the queue.
#include <cstring>
#include <mutex>

#include "bodies.hpp"
#include "server.hpp"

namespace spiral {

// ------------------------------------------------------------------------------------------------ recorder (host)
static thread_local Program* g_rec = nullptr;
static thread_local int g_rec_group = 0;  // > 0: inside program_group_begin .. _end

Program::~Program() {
  if (dev) (void)hipFree(dev);
  if (ctr) (void)hipFree(ctr);
}
void program_begin(Program& p) {
  p.phases.clear();
  g_rec = &p;
  g_rec_group = 0;
}
void program_end() {
  if (g_rec && !g_rec->phases.empty()) g_rec->phases.back().no_barrier = 0;
  g_rec = nullptr;
  g_rec_group = 0;
}
bool program_recording() { return g_rec != nullptr; }
Program* program_current() { return g_rec; }
void program_group_begin() {
  if (g_rec) g_rec_group++;
}
void program_group_end() {
  if (!g_rec || g_rec_group == 0) return;
  if (--g_rec_group == 0 && !g_rec->phases.empty()) g_rec->phases.back().no_barrier = 0;  // the group's last phase ends with a barrier
}
static Phase* new_phase(int kind, long units) {
  if (!g_rec || units <= 0) return nullptr;
  Phase ph;
  memset(static_cast<void*>(&ph), 0, sizeof(ph));
  ph.kind = kind;
  ph.units = (int)units;
  ph.no_barrier = g_rec_group > 0 ? 1 : 0;
  g_rec->phases.push_back(ph);
  return &g_rec->phases.back();
}
bool program_record(const FwdDesc& d) {
  if (!g_rec) return false;
  if (Phase* p = new_phase(PH_FWD, (long)d.n_out * 2)) p->fwd[0] = d;
  return true;
}
bool program_record(const FwdDesc& a, const FwdDesc& b, const FwdDesc& c) {
  if (!g_rec) return false;
  if (Phase* p = new_phase(PH_FWD3, ((long)a.n_out + b.n_out + c.n_out) * 2)) {
    p->fwd[0] = a;
    p->fwd[1] = b;
    p->fwd[2] = c;
  }
  return true;
}
bool program_record(const InvDesc& d) {
  if (!g_rec) return false;
  if (Phase* p = new_phase(PH_INV, (long)d.n_polys + (d.scal ? 2L * d.n_scalar_only : 0L))) p->inv = d;
  return true;
}
bool program_record(const MacDesc& d) {
  if (!g_rec) return false;
  if (Phase* p = new_phase(PH_MAC, (long)d.batch_inner * d.batch_outer * (2 * N / 256))) p->mac[0] = d;
  return true;
}
bool program_record(const MacDesc& a, const MacDesc& b) {
  if (!g_rec) return false;
  if (Phase* p = new_phase(PH_MAC2, ((long)a.batch_inner + b.batch_inner) * (2 * N / 256))) {
    p->mac[0] = a;
    p->mac[1] = b;
  }
  return true;
}
bool program_record(const CopyPolysDesc& d) {
  if (!g_rec) return false;
  if (Phase* p = new_phase(PH_COPY_POLYS, (long)d.batch * d.R * (2 * N / 256))) p->copy_polys = d;
  return true;
}
bool program_record(const FoldingNegDesc& d) {
  if (!g_rec) return false;
  if (Phase* p = new_phase(PH_FOLDING_NEG, (long)(2 * N / 256) * 2 * d.two_t * d.nu2)) p->folding_neg = d;
  return true;
}
bool program_record(const ReorientDesc& d) {
  if (!g_rec) return false;
  if (Phase* p = new_phase(PH_REORIENT, (long)((d.dim0 + 31) / 32) * (N / 32) * 2)) p->reorient = d;
  return true;
}
bool program_record(const MatsToWaveDesc& d) {
  if (!g_rec) return false;
  if (Phase* p = new_phase(PH_MATS_TO_WAVE, (long)((d.n_words + 255) / 256))) p->mats_to_wave = d;
  return true;
}
bool program_record(const AddPolyIntoDesc& d) {
  if (!g_rec) return false;
  if (Phase* p = new_phase(PH_ADD_POLY_INTO, (long)d.batch * (2 * N / 256))) p->add_poly_into = d;
  return true;
}
bool program_record(const EncodeDesc& d) {
  if (!g_rec) return false;
  const long total = (long)d.instances * ((long)d.n * N + (long)d.n * d.n * N);
  if (Phase* p = new_phase(PH_ENCODE, (total + 255) / 256)) p->encode = d;
  return true;
}
bool program_record(const CopyWordsDesc& d) {
  if (!g_rec) return false;
  if (Phase* p = new_phase(PH_COPY_WORDS, (long)((d.n_words + 1023) / 1024))) p->copy_words = d;
  return true;
}
bool program_record_fill_zero(u32* dst, size_t n_words) {
  if (!g_rec) return false;
  if (Phase* p = new_phase(PH_FILL_ZERO, (long)((n_words + 1023) / 1024))) p->copy_words = CopyWordsDesc{dst, nullptr, n_words};
  return true;
}

// ------------------------------------------------------------------------------------------------ kernel
constexpr long PROGRAM_SPIN_LIMIT = 2000000;  // a turn = s_sleep 2 + one agent-scope load (~1-2 us): a few seconds

__device__ __forceinline__ void program_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();  // every wave of the workgroup has issued its stores of the phase and waited for them (vmcnt 0)
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // L2 write-back: visible to the other XCDs
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > PROGRAM_SPIN_LIMIT) __builtin_trap();  // a lost workgroup must not hang the queue for ever
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // L1 / L2 invalidate: the other workgroups' results are read from memory
  }
  __syncthreads();
}

// The exchange buffers of the transforms live at module scope: a pointer PARAMETER of a non-inlined function is a generic
// pointer and every access through it a flat_load / flat_store with an aperture check instead of a ds_ instruction.
__shared__ __attribute__((aligned(16))) u32 g_program_lds[2 * LDS_WORDS];

// One function per phase kind, NOT inlined: inlined into one kernel body the register allocator sees every kind's loop
// invariants at once and spills (528 bytes of scratch per lane at 128 VGPRs); as calls, each kind gets the file to itself.
#define SP_PHASE_FN __device__ __noinline__ static void
SP_PHASE_FN phase_fwd(const DevTables& T, const Phase* ph) {
  u32* const lds = g_program_lds;
  const int grid = gridDim.x, units = ph->units;
  for (int u = blockIdx.x; u < units; u += grid) {
    ntt_fwd_body(T, ph->fwd[0], u >> 1, u & 1, lds, lds + LDS_WORDS);
    __syncthreads();  // the LDS buffers are reused by the next unit
  }
}
SP_PHASE_FN phase_fwd3(const DevTables& T, const Phase* ph) {
  u32* const lds = g_program_lds;
  const int grid = gridDim.x, units = ph->units;
  const int n0 = ph->fwd[0].n_out, n1 = ph->fwd[1].n_out;
  for (int u = blockIdx.x; u < units; u += grid) {
    const int o = u >> 1, c = u & 1;
    const int which = o < n0 ? 0 : o < n0 + n1 ? 1 : 2;
    ntt_fwd_body(T, ph->fwd[which], o - (which == 0 ? 0 : which == 1 ? n0 : n0 + n1), c, lds, lds + LDS_WORDS);
    __syncthreads();
  }
}
SP_PHASE_FN phase_inv(const DevTables& T, const Phase* ph) {
  u32* const lds = g_program_lds;
  const int grid = gridDim.x, units = ph->units;
  for (int u = blockIdx.x; u < units; u += grid) {
    ntt_inv_body(T, ph->inv, u, lds, lds + LDS_WORDS);
    __syncthreads();
  }
}
SP_PHASE_FN phase_mac(const DevTables& T, const Phase* ph) {
  const int grid = gridDim.x, units = ph->units;
  const int per = ph->mac[0].batch_inner * 16;
  for (int u = blockIdx.x; u < units; u += grid) {
    const int outer = u / per, rem = u - outer * per;
    mac_body<14>(T, ph->mac[0], rem >> 4, outer, rem & 15);
  }
}
SP_PHASE_FN phase_mac2(const DevTables& T, const Phase* ph) {
  const int grid = gridDim.x, units = ph->units;
  const int b0 = ph->mac[0].batch_inner;
  for (int u = blockIdx.x; u < units; u += grid) {
    const int y = u >> 4, which = y < b0 ? 0 : 1;
    mac_body<14>(T, ph->mac[which], which ? y - b0 : y, 0, u & 15);
  }
}
SP_PHASE_FN phase_small(const DevTables& T, const Phase* ph) {  // the elementwise kinds
  u32* const lds = g_program_lds;
  const int grid = gridDim.x, units = ph->units;
  switch (ph->kind) {
    case PH_COPY_POLYS:
      for (int u = blockIdx.x; u < units; u += grid) copy_polys_body(ph->copy_polys, u >> 4, u & 15);
      break;
    case PH_FOLDING_NEG: {
      const int rows = 2 * ph->folding_neg.two_t;
      for (int u = blockIdx.x; u < units; u += grid) {
        const int ych = u & 15, rest = u >> 4;
        folding_neg_body(T, ph->folding_neg, ych, rest % rows, rest / rows);
      }
      break;
    }
    case PH_REORIENT: {
      const int nbx = (ph->reorient.dim0 + 31) / 32;
      for (int u = blockIdx.x; u < units; u += grid) {
        const int bx = u % nbx, rest = u / nbx;
        reorient_body(ph->reorient, bx, rest % (N / 32), rest / (N / 32), reinterpret_cast<u64*>(lds));
        __syncthreads();
      }
      break;
    }
    case PH_MATS_TO_WAVE:
      for (int u = blockIdx.x; u < units; u += grid) mats_to_wave_body(ph->mats_to_wave, (size_t)u);
      break;
    case PH_ADD_POLY_INTO:
      for (int u = blockIdx.x; u < units; u += grid)
        add_poly_into_body(T, ph->add_poly_into.dst, ph->add_poly_into.idx, ph->add_poly_into.src, u >> 4, u & 15);
      break;
    case PH_ENCODE:
      for (int u = blockIdx.x; u < units; u += grid) encode_body(ph->encode, u);
      break;
    case PH_COPY_WORDS:
      for (int u = blockIdx.x; u < units; u += grid)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const size_t i = (size_t)u * 1024 + k * 256 + threadIdx.x;
          if (i < ph->copy_words.n_words) ph->copy_words.dst[i] = ph->copy_words.src[i];
        }
      break;
    case PH_FILL_ZERO:
      for (int u = blockIdx.x; u < units; u += grid)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const size_t i = (size_t)u * 1024 + k * 256 + threadIdx.x;
          if (i < ph->copy_words.n_words) ph->copy_words.dst[i] = 0u;
        }
      break;
    default: break;
  }
}
#undef SP_PHASE_FN

__global__ __launch_bounds__(256, 4) void k_program(DevTables T, const Phase* __restrict__ phases, int n_phases, unsigned* ctr,
                                                    unsigned ctr_base) {
  unsigned barriers = 0;
  for (int pi = 0; pi < n_phases; pi++) {
    const Phase* ph = phases + pi;
    switch (ph->kind) {
      case PH_FWD: phase_fwd(T, ph); break;
      case PH_FWD3: phase_fwd3(T, ph); break;
      case PH_INV: phase_inv(T, ph); break;
      case PH_MAC: phase_mac(T, ph); break;
      case PH_MAC2: phase_mac2(T, ph); break;
      default: phase_small(T, ph); break;
    }
    if (pi + 1 < n_phases && !ph->no_barrier) {
      barriers++;
      program_barrier(ctr, ctr_base + barriers * gridDim.x);
    }
  }
}

// ------------------------------------------------------------------------------------------------ launch
// Every program launch of the process waits for the previous one (whatever stream it ran on): two persistent grids that
// each hold some of a CU's slots and spin for the rest would never finish.  One event per launch, recycled from a ring.
namespace {
std::mutex g_chain_mu;
constexpr int CHAIN_RING = 64;
struct ChainState {
  int device = -1;
  hipEvent_t ev[CHAIN_RING] = {};
  int next = 0;
  hipEvent_t last = nullptr;
};
ChainState g_chain[16];
ChainState& chain_for_device(int dev) {
  for (auto& c : g_chain)
    if (c.device == dev) return c;
  for (auto& c : g_chain)
    if (c.device < 0) {
      c.device = dev;
      return c;
    }
  return g_chain[0];
}
}  // namespace

void program_launch(const DevTables& T, Program& p, hipStream_t s, u64 path_bits) {
  const size_t n = p.phases.size();
  if (n == 0) return;
  if (!p.ctr) {
    if (hipMalloc((void**)&p.ctr, 256) != hipSuccess) throw OomError("hipMalloc of a program's barrier counter failed");
    (void)hipMemset(p.ctr, 0, 256);
    (void)hipDeviceSynchronize();
    p.ctr_next = 0;
  }
  const bool same = p.on_device.size() == n && memcmp(p.on_device.data(), p.phases.data(), n * sizeof(Phase)) == 0;
  if (!same) {
    if (p.dev_cap < n) {
      if (p.dev) {
        (void)hipStreamSynchronize(s);  // a previous launch on this stream may still read the old buffer
        (void)hipFree(p.dev);
        p.dev = nullptr;
      }
      const size_t cap = std::max<size_t>(n, 64);
      if (hipMalloc((void**)&p.dev, cap * sizeof(Phase)) != hipSuccess) throw OomError("hipMalloc of a program's phase list failed");
      p.dev_cap = cap;
    }
    // stream-ordered after the previous launch of this program (same workspace, same stream); the source is pageable, so
    // the runtime has staged it before this returns
    if (hipMemcpyAsync(p.dev, p.phases.data(), n * sizeof(Phase), hipMemcpyHostToDevice, s) != hipSuccess)
      throw HipError("upload of a program's phase list failed");
    p.on_device = p.phases;
  }
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  {
    static thread_local int cached_dev = -1, cached_cus = 256;
    if (cached_dev != dev) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cached_cus = prop.multiProcessorCount;
      cached_dev = dev;
    }
    cus = cached_cus;
  }
  long wgs = tunable("program_wgs", 2);
  wgs = std::max(1L, std::min(4L, wgs));
  long max_units = 1;
  unsigned barriers = 0;
  for (size_t i = 0; i < n; i++) {
    max_units = std::max<long>(max_units, p.phases[i].units);
    if (i + 1 < n && !p.phases[i].no_barrier) barriers++;
  }
  const unsigned grid = (unsigned)std::max(1L, std::min((long)cus * wgs, max_units));
  std::lock_guard<std::mutex> lk(g_chain_mu);
  ChainState& ch = chain_for_device(dev);
  if (ch.last && hipStreamWaitEvent(s, ch.last, 0) != hipSuccess) throw HipError("hipStreamWaitEvent failed");
  hipLaunchKernelGGL(k_program, dim3(grid), dim3(256), 0, s, T, p.dev, (int)n, p.ctr, p.ctr_next);
  launched(path_bits, "k_program");
  p.ctr_next += barriers * grid;
  hipEvent_t& e = ch.ev[ch.next];
  if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) throw HipError("hipEventCreate failed");
  if (hipEventRecord(e, s) != hipSuccess) throw HipError("hipEventRecord failed");
  ch.last = e;
  ch.next = (ch.next + 1) % CHAIN_RING;
}

}  // namespace spiral
