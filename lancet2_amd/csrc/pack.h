// Packed result records of the host route (pack.hip).
#pragma once
#include "ma_internal.h"

namespace ma {

enum PackKind : u32 { PK_VAR = 0, PK_POOL = 1, PK_HAP_BASES = 2, PK_HAP_RUNS = 3 };

struct PackSeg {
  const u8* src;    // the array's first window (device pointer in the kernels, the caller's host pointer in unpack_records)
  u32 win_stride;   // bytes per window
  u32 unit_bytes;   // PK_VAR: bytes per variant; PK_HAP_*: bytes per haplotype slot
  u32 kind;
};

struct PackArgs {
  // what says how much of a window's slices is in use
  const u32* win_status;
  const u32* win_ncomp;
  const u32* comp_hap0;
  const u32* comp_nhaps;
  const u32* hap_len;
  const u32* hap_nruns;
  const u32* win_nvars;
  const u32* var_ref_off;
  const u32* var_ref_len;
  const u32* var_nalts;
  const u32* alt_off;
  const u32* alt_len;
  u32 MC, MH, MV, MA, MP;
  u32 nseg;
  PackSeg seg[24];
};

// whole-array copies of k_pack_dense (sizes in bytes, multiples of 4)
struct DenseCopy {
  const void* src;
  void* dst;
  u32 bytes;
};
struct DenseCopies {
  DenseCopy c[20];
  u32 n;
};

// The landing area is pinned HOST memory that the kernels write themselves (the records cross PCIe as the stores of
// k_pack_copy / k_pack_dense): no copy commands, and one wait for the whole of a lane's results.
// aux_dev: [2 n + 1] u32 -- per window {record offset in 4-byte words, allele pool bytes in use}, then the total words
int launch_pack_sizes(ma_ctx* ctx, PackArgs const& A, int n, u32* aux_dev);
int launch_pack_dense(ma_ctx* ctx, DenseCopies const& C);
// records to `out` if they fit in cap_bytes (aux[2 n] * 4 <= cap_bytes: the caller checks the total afterwards)
int launch_pack_records(ma_ctx* ctx, PackArgs const& A, int n, const u32* aux_dev, u8* out, size_t cap_bytes);
void unpack_records(PackArgs const& H, const u32* aux, const u8* packed, int n);

}  // namespace ma
