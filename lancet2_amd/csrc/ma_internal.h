// Internal definitions of the MI355X microassembly engine (product code; never includes oracle/).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/microasm.h"

namespace ma {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i8 = int8_t;
using i16 = int16_t;
using i32 = int32_t;
using i64 = int64_t;
using f64 = double;

// node-identity hash shared with the oracle (oracle/common.hpp: HashStr64)
constexpr u64 kHashP = 0x9E3779B97F4A7C15ULL;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

struct KernelTimer {
  const char* name;
  hipEvent_t beg, end;
};

}  // namespace ma

struct ma_ctx {
  ma_params_t prm;
  int device = 0;
  int memspace = MA_MEM_HOST;
  hipStream_t stream = nullptr;
  std::string err;
  // staging for MA_MEM_HOST
  ma::DevBuf in_stage[10];
  std::vector<ma::DevBuf> out_stage;
  // per-stage workspaces (grow-only, reused across calls)
  ma::DevBuf ws_build, ws_nodes, ws_clean, ws_poa, ws_aln, ws_misc;
  // kernel timing
  std::vector<ma::KernelTimer> timers;
  size_t timers_used = 0;
  bool timing = true;
  bool accumulate = false;

  void tic(const char* name);
  void toc();
};

#define MA_HIP(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t _e = (call);                                                            \
    if (_e != hipSuccess) {                                                            \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(_e);                  \
      return MA_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

#define MA_TRY_RC(expr)            \
  do {                             \
    int _rc = (expr);              \
    if (_rc != MA_OK) return _rc;  \
  } while (0)

namespace ma {

// device-side views (all pointers are device pointers)
struct DBatch {
  int n_windows;
  i64 n_reads;
  const u8* ref_bases;
  const u32* ref_off;
  const u32* read_win_off;
  const u64* read_off;
  const u8* read_bases;
  const u8* read_quals;
  const u32* read_qname_id;
  const u8* read_sample;
  const u8* read_flags;
  const i32* read_hint;  // may be null
};

int launch_gate(ma_ctx* ctx, const DBatch& b, u32* max_approx, u32* max_exact);
int launch_assemble(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& o, const u32* gate_approx);
int launch_msa(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& o);
int launch_genotype(ma_ctx* ctx, const DBatch& b, const ma_asm_out_t& a, const ma_var_out_t& v,
                    const ma_geno_out_t& o);

}  // namespace ma
