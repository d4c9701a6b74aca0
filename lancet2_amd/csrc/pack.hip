// Compact result transfer of the host route (MA_MEM_HOST).  The caller's output buffers are fixed-stride arrays sized for
// the caps (max_haps x max_hap_len bases, max_vars records, ...): 95 KB per window of which a typical window uses 5.  Copying
// them whole was a third of a host batch.  Here the used prefix of every array's window slice is gathered into one packed
// record per window (k_pack_size -> scan -> k_pack_copy), the records cross PCIe in one copy, and the host scatters them into
// the caller's arrays (unpack_records).  Bytes the engine never wrote are not transferred, and stay untouched on the host.
#include <algorithm>
#include <cstring>

#include "ma_internal.h"
#include "pack.h"

namespace ma {

namespace {

__device__ __forceinline__ u32 pad4(u32 b) { return (b + 3u) & ~3u; }

struct WinUse {
  u32 nhap;   // haplotype slots in use
  u32 nvar;
  u32 pool;   // allele pool bytes in use
};

__device__ WinUse window_use(PackArgs const& A, int w) {
  WinUse u{0, 0, 0};
  {  // (a window without an ALT haplotype may still report components with their REF haplotype)
    u32 const nc = min(A.win_ncomp[w], A.MC);
    for (u32 c = 0; c < nc; ++c) {
      size_t const ci = static_cast<size_t>(w) * A.MC + c;
      u.nhap = max(u.nhap, A.comp_hap0[ci] + A.comp_nhaps[ci]);
    }
    u.nhap = min(u.nhap, A.MH);
    u.nvar = min(A.win_nvars[w], A.MV);
  }
  for (u32 x = 0; x < u.nvar; ++x) {
    size_t const vi = static_cast<size_t>(w) * A.MV + x;
    u.pool = max(u.pool, A.var_ref_off[vi] + A.var_ref_len[vi]);
    u32 const na = min(A.var_nalts[vi], A.MA);
    for (u32 a = 0; a < na; ++a) u.pool = max(u.pool, A.alt_off[vi * A.MA + a] + A.alt_len[vi * A.MA + a]);
  }
  u.pool = min(u.pool, A.MP);
  return u;
}

// bytes of window w's record; aux[w] = {words (filled in by the scan: offset), pool bytes}
__global__ void k_pack_size(PackArgs A, u32* aux, int n) {
  int const w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  WinUse const u = window_use(A, w);
  u32 bytes = 0;
  for (u32 s = 0; s < A.nseg; ++s) {
    PackSeg const& sg = A.seg[s];
    if (sg.kind == PK_VAR) {
      bytes += pad4(u.nvar * sg.unit_bytes);
    } else if (sg.kind == PK_POOL) {
      bytes += pad4(u.pool);
    } else {
      for (u32 h = 0; h < u.nhap; ++h) {
        size_t const hi = static_cast<size_t>(w) * A.MH + h;
        bytes += pad4(sg.kind == PK_HAP_BASES ? A.hap_len[hi] : A.hap_nruns[hi] * 8u);
      }
    }
  }
  aux[2 * w] = bytes / 4u;
  aux[2 * w + 1] = u.pool;
}

// exclusive scan of aux[2 w] over the windows (one workgroup; n is a few thousand), total -> aux[2 n]
__global__ __launch_bounds__(1024) void k_pack_scan(u32* aux, int n) {
  __shared__ u32 sh[1024];
  __shared__ u32 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int const i = base + threadIdx.x;
    u32 const v = i < n ? aux[2 * i] : 0u;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      u32 const x = threadIdx.x >= static_cast<u32>(d) ? sh[threadIdx.x - d] : 0u;
      __syncthreads();
      sh[threadIdx.x] += x;
      __syncthreads();
    }
    u32 const incl = sh[threadIdx.x];
    if (i < n) aux[2 * i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) aux[2 * n] = carry;
}

__device__ __forceinline__ void copy_bytes(u8* dst, const u8* src, u32 bytes, int lane) {
  // (sources are byte arrays at arbitrary strides: byte loads; destinations are 4-byte aligned)
  if ((reinterpret_cast<uintptr_t>(src) & 3u) == 0) {
    u32 const words = bytes / 4u;
    for (u32 i = lane; i < words; i += 64) reinterpret_cast<u32*>(dst)[i] = reinterpret_cast<const u32*>(src)[i];
    for (u32 i = words * 4u + lane; i < bytes; i += 64) dst[i] = src[i];
  } else {
    for (u32 i = lane; i < bytes; i += 64) dst[i] = src[i];
  }
}

__global__ __launch_bounds__(256) void k_pack_copy(PackArgs A, const u32* aux, u8* out, int n, u32 cap_words) {
  int const w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= n) return;
  if (aux[2 * n] > cap_words) return;  // the landing area is too small: the host sees the total, grows it and asks again
  WinUse const u = window_use(A, w);
  u8* dst = out + static_cast<size_t>(aux[2 * w]) * 4u;
  for (u32 s = 0; s < A.nseg; ++s) {
    PackSeg const& sg = A.seg[s];
    const u8* src = sg.src + static_cast<size_t>(w) * sg.win_stride;
    if (sg.kind == PK_VAR || sg.kind == PK_POOL) {
      u32 const bytes = sg.kind == PK_VAR ? u.nvar * sg.unit_bytes : u.pool;
      copy_bytes(dst, src, bytes, lane);
      dst += pad4(bytes);
    } else {
      for (u32 h = 0; h < u.nhap; ++h) {
        size_t const hi = static_cast<size_t>(w) * A.MH + h;
        u32 const bytes = sg.kind == PK_HAP_BASES ? A.hap_len[hi] : A.hap_nruns[hi] * 8u;
        copy_bytes(dst, src + static_cast<size_t>(h) * sg.unit_bytes, bytes, lane);
        dst += pad4(bytes);
      }
    }
  }
}

// whole arrays (the small per-window ones, and aux itself) into the landing area: 4-byte words, one workgroup row per array
__global__ __launch_bounds__(256) void k_pack_dense(DenseCopies C) {
  DenseCopy const& c = C.c[blockIdx.y];
  u32 const words = c.bytes / 4u;
  const u32* src = reinterpret_cast<const u32*>(c.src);
  u32* dst = reinterpret_cast<u32*>(c.dst);
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < words; i += gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

int launch_pack_sizes(ma_ctx* ctx, PackArgs const& A, int n, u32* aux_dev) {
  hipLaunchKernelGGL(k_pack_size, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, A, aux_dev, n);
  hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, ctx->stream, aux_dev, n);
  MA_HIP(ctx, hipGetLastError());
  return MA_OK;
}

int launch_pack_dense(ma_ctx* ctx, DenseCopies const& C) {
  if (C.n) hipLaunchKernelGGL(k_pack_dense, dim3(64, C.n), dim3(256), 0, ctx->stream, C);
  MA_HIP(ctx, hipGetLastError());
  return MA_OK;
}

int launch_pack_records(ma_ctx* ctx, PackArgs const& A, int n, const u32* aux_dev, u8* out, size_t cap_bytes) {
  u32 const cap_words = static_cast<u32>(std::min<size_t>(cap_bytes / 4u, 0xFFFFFFFFu));
  hipLaunchKernelGGL(k_pack_copy, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, A, aux_dev, out, n, cap_words);
  MA_HIP(ctx, hipGetLastError());
  return MA_OK;
}

// Host side: the same walk, from the packed records into the caller's arrays.  `H` holds HOST pointers (segment sources
// = the caller's arrays, already advanced to the lane's first window) and the host copies of the small dense arrays.
void unpack_records(PackArgs const& H, const u32* aux, const u8* packed, int n) {
  for (int w = 0; w < n; ++w) {
    u32 nhap = 0, nvar = 0;
    {
      u32 const nc = std::min(H.win_ncomp[w], H.MC);
      for (u32 c = 0; c < nc; ++c) {
        size_t const ci = static_cast<size_t>(w) * H.MC + c;
        nhap = std::max(nhap, H.comp_hap0[ci] + H.comp_nhaps[ci]);
      }
      nhap = std::min(nhap, H.MH);
      nvar = std::min(H.win_nvars[w], H.MV);
    }
    u32 const pool = aux[2 * w + 1];
    const u8* src = packed + static_cast<size_t>(aux[2 * w]) * 4u;
    auto p4 = [](u32 b) { return (b + 3u) & ~3u; };
    for (u32 s = 0; s < H.nseg; ++s) {
      PackSeg const& sg = H.seg[s];
      u8* dst = const_cast<u8*>(sg.src) + static_cast<size_t>(w) * sg.win_stride;
      if (sg.kind == PK_VAR || sg.kind == PK_POOL) {
        u32 const bytes = sg.kind == PK_VAR ? nvar * sg.unit_bytes : pool;
        if (bytes) std::memcpy(dst, src, bytes);
        src += p4(bytes);
      } else {
        for (u32 h = 0; h < nhap; ++h) {
          size_t const hi = static_cast<size_t>(w) * H.MH + h;
          u32 const bytes = sg.kind == PK_HAP_BASES ? H.hap_len[hi] : H.hap_nruns[hi] * 8u;
          if (bytes) std::memcpy(dst + static_cast<size_t>(h) * sg.unit_bytes, src, bytes);
          src += p4(bytes);
        }
      }
    }
  }
}

}  // namespace ma
