// K5s: 3x3 / stride-1 / pad-1 w4a8 convolution as an implicit GEMM whose A operand is staged ONCE per channel chunk.
//
// The 128x128 kernel (conv_igemm.hip: k_conv_dma) re-reads the activation tile for each of the 9 taps and is bound by
// the L2 -> LDS stream (~27 B/clk/CU sustained against 64 B/clk/CU needed at the MFMA peak).  Here
//   * the K order is (channel chunk, tap): for one 64-channel chunk the block brings the (rows + halo) x (W + 2) pixel
//     "slab" that its 256 output pixels touch into LDS once (<= 512 rows x 64 B), and all nine taps read their A
//     fragments from it at a tap-dependent row offset -- A traffic per K-step drops ~8x and the padded border is part
//     of the slab (rows of the byte z_a - 128 from the handle's pad table: a real zero under the activation quantizer);
//   * the tile is 256 pixels x (64 * WN) channels on 8 waves (4 along M x 2 along N, each 64 x 32*WN), WN = 5 gives
//     320-wide tiles: no column waste at Cout = 320 / 640 / 960 / 1280 ..., B (weight) bytes per MFMA halve against 128^2;
//   * both operands travel global -> LDS by LDS-DMA with the 16-byte-slot XOR swizzle of the other kernels (on the
//     source address and on the fragment reads), the slab double-buffered per chunk, the weights in three K-step
//     stages, counted vmcnt and one raw s_barrier per K-step.
// int32 accumulation is exact, so the result equals k_conv_dma's bit for bit whatever the K order; the epilogue
// (dequantise, + bias, + temb row, + residual, GroupNorm statistics in the one summation order every tile shape
// uses) stages 32 tile rows at a time through LDS and writes whole rows.
#include "conv_common.hpp"
#include <type_traits>

namespace {

struct SlabP {
  ConvP p;
  int HW;         // pixels per image
  int imgs;       // images per 256-pixel tile (> 1 when an image has fewer than 256 pixels)
  int SW;         // slab row width = W + 2
  int SI;         // slab rows per image (imgs > 1); imgs == 1: rows of the whole slab
  int slab_rows;  // rows of the slab that carry pixels (<= 512)
};

constexpr int SLAB_CAP = 512;                 // rows per slab buffer
constexpr int SLAB_BYTES = SLAB_CAP * 64;     // 32 KiB, a power of two: the two buffers toggle by XOR on the offset

template <int WN>
__host__ __device__ constexpr int slab_lds_bytes() {
  constexpr int BN = 64 * WN;
  constexpr int main_ = 2 * SLAB_BYTES + 3 * BN * 64;
  constexpr int epi = 32 * (BN + 4) * 4 + 32 * BN * 8;
  return main_ > epi ? main_ : epi;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WN>
__global__ __launch_bounds__(512) void k_conv3_slab(SlabP sp) {
  constexpr int BM = 256, BN = 64 * WN;
  constexpr int BST = BN * 64;                 // bytes of one weight K-step stage
  constexpr int BOFF = 2 * SLAB_BYTES;
  constexpr int NBP = BN / 16;                 // 1-KiB weight pieces per K-step
  __shared__ __attribute__((aligned(1024))) unsigned char lds[slab_lds_bytes<WN>()];

  const ConvP& p = sp.p;
  const tfmq_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int bid = xcd_tile_id();
  const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int W = p.Wv, H = p.Hv, SW = sp.SW;      // the (virtual, when the nearest-2x upsample is fused) input image
  const int ups = d.up2x ? 1 : 0;

  const float2 aqp = load_qparam(d.aq);

  // ---- slab geometry of this tile
  const int b0 = m0 / sp.HW;
  const int y0 = sp.imgs == 1 ? (m0 - b0 * sp.HW) / W : 0;       // first image row of the tile
  // A fragments: tile row of this lane in M-tile i -> slab row of the top-left pixel of its 3x3 window
  int srow0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ml = wm * 64 + i * 32 + (lane & 31);
    if (sp.imgs == 1) {
      const int yr = ml / W;
      srow0[i] = yr * SW + (ml - yr * W);
    } else {
      const int im = ml / sp.HW, q = ml - im * sp.HW, y = q / W;
      srow0[i] = im * sp.SI + y * SW + (q - y * W);
    }
  }
  // slab DMA sources: piece it*8 + wid, 16 rows each, this lane's row = piece*16 + lane/4
  const unsigned char* xb = static_cast<const unsigned char*>(d.x);
  const int dcol = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;        // swizzle on the SOURCE: (row >> 2) & 3 == (lane >> 4) & 3
  int s_off[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int j = (it * 8 + wid) * 16 + (lane >> 2);
    int off = -1;
    if (j < sp.slab_rows) {
      const int k = sp.imgs == 1 ? 0 : j / sp.SI;
      const int rem = j - k * sp.SI;
      const int sy = rem / SW, sx = rem - sy * SW;
      const int b = b0 + k, y = (sp.imgs == 1 ? y0 : 0) - 1 + sy, x = sx - 1;
      // fused upsample: the slab holds the UPSAMPLED pixels -- virtual (y, x) reads input (y >> 1, x >> 1), four slab rows
      // per input pixel, all but the first from L2
      if (b < d.B && y >= 0 && y < H && x >= 0 && x < W) off = ((b * d.H + (y >> ups)) * d.W + (x >> ups)) * d.Cin;
    }
    s_off[it] = off;
  }
  const int za = static_cast<int>(aqp.y);
  const unsigned char* padp = p.pad_table + (static_cast<unsigned>(za - 128) & 0xffu) * 64 + dcol;
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));

  auto issue_slab = [&](int it, int c, int buf) {
    const unsigned char* src = s_off[it] >= 0 ? xb + static_cast<size_t>(static_cast<unsigned>(s_off[it])) + c * 64 + dcol : padp;
    glds16(src, lds0 + buf * SLAB_BYTES + __builtin_amdgcn_readfirstlane((it * 8 + wid) * 1024));
  };

  // weight DMA sources: pieces wid, wid + 8, wid + 16 (< NBP)
  const unsigned char* b_ptr[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int piece = wid + 8 * k;
    int n = n0 + piece * 16 + (lane >> 2);
    n = n < p.cout_pad ? n : p.cout_pad - 1;
    b_ptr[k] = static_cast<const unsigned char*>(d.w) + (static_cast<size_t>(n / 32) * p.nsteps * 32 + (n % 32)) * 64 + dcol;
  }

  // fragment read offsets
  const int b_rel = (wn * WN * 32 + (lane & 31)) * 64 + ((h ^ ((lane >> 2) & 3)) << 4);
  int slab_toggle = 0;

  v16i acc[2][WN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

  auto kloop = [&](auto bch_tag) {
    constexpr int B_CH = decltype(bch_tag)::value;
    auto issue_b = [&](int c, int tap, int stage) {
      const size_t boff = static_cast<size_t>(tap * p.chunks + c) * 2048;
#pragma unroll
      for (int k = 0; k < B_CH; ++k)
        glds16(b_ptr[k] + boff, lds0 + BOFF + stage * BST + __builtin_amdgcn_readfirstlane((wid + 8 * k) * 1024));
    };
    auto step = [&](int c, auto tap_tag) {
      constexpr int TAP = decltype(tap_tag)::value;
      // slab pieces issued one / two iterations ago (taps 0..3 carry one each) are younger than this step's weights
      constexpr int X = ((TAP >= 1 && TAP <= 4) ? 1 : 0) + ((TAP >= 2 && TAP <= 5) ? 1 : 0);
      const int pos = c * 9 + TAP;
      if (pos + 1 < p.nsteps) wait_vmcnt<B_CH + X>();
      else wait_vmcnt<0>();
      asm volatile("s_barrier" ::: "memory");
      auto issue_next = [&]() {
        if (pos + 2 < p.nsteps) {
          if constexpr (TAP + 2 < 9) issue_b(c, TAP + 2, (TAP + 2) % 3);
          else issue_b(c + 1, TAP + 2 - 9, (TAP + 2) % 3);
        }
        if constexpr (TAP < 4) {
          // next chunk's slab into the idle buffer; the last chunk re-stages chunk 0 there (never read) so that the
          // counted waits see the same number of loads in flight in every chunk
          issue_slab(TAP, c + 1 < p.chunks ? c + 1 : 0, ((c + 1) & 1));
        }
      };
      constexpr int KH = TAP / 3, KW = TAP % 3;
      const int toff = KH * SW + KW;
      int a_rel[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int srow = srow0[i] + toff;
        a_rel[i] = ((srow << 6) + (((h ^ (srow >> 2)) & 3) << 4)) ^ slab_toggle;
      }
      const unsigned char* sb = lds + BOFF + (TAP % 3) * BST;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        v4i af[2], bf[WN];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const v4i*>(lds + (a_rel[i] ^ (ks << 5)));
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[j] = *reinterpret_cast<const v4i*>(sb + ((b_rel ^ (ks << 5)) + j * 2048));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        // the DMA issue of the next K-steps (SALU M0 moves + VMEM, ~100 clk a piece) sits behind the first half's MFMAs:
        // the matrix pipe works through them while the wave issues the loads, instead of idling right after the barrier
        if (ks == 0) issue_next();
      }
    };
#pragma unroll
    for (int it = 0; it < 4; ++it) issue_slab(it, 0, 0);
    issue_b(0, 0, 0);
    issue_b(0, 1, 1);
    for (int c = 0; c < p.chunks; ++c) {
      step(c, std::integral_constant<int, 0>{});
      step(c, std::integral_constant<int, 1>{});
      step(c, std::integral_constant<int, 2>{});
      step(c, std::integral_constant<int, 3>{});
      step(c, std::integral_constant<int, 4>{});
      step(c, std::integral_constant<int, 5>{});
      step(c, std::integral_constant<int, 6>{});
      step(c, std::integral_constant<int, 7>{});
      step(c, std::integral_constant<int, 8>{});
      slab_toggle ^= SLAB_BYTES;
    }
  };
  if (NBP % 8 != 0 && wid < NBP % 8) kloop(std::integral_constant<int, NBP / 8 + 1>{});
  else kloop(std::integral_constant<int, NBP / 8>{});

  // ================================================================================ epilogue
  constexpr int LDO = BN + 4, TPR = BN / 4;
  float* ldsO = reinterpret_cast<float*>(lds);                          // [32][LDO]
  float2* ldsP = reinterpret_cast<float2*>(lds + 32 * LDO * 4);         // [32 eight-row groups][BN] (sum, sum of squares)
  const int hw = sp.HW;
  const float* rowadd = d.rowadd;
  if (rowadd && d.rowadd_step) rowadd += static_cast<size_t>(*d.rowadd_step) * d.rowadd_step_stride;
  const int seg = d.stats ? d.stats_seg : 0;
  const bool q8 = d.out_mode == TFMQ_OUT_Q8, o16 = d.out_mode == TFMQ_OUT_F16;
  const bool q16 = (d.Cout & 15) == 0, h8 = ((d.Cout | d.ldy | d.y_coff) & 7) == 0;     // 16-byte item paths of the store pass
  float2 oqp = make_float2(1.0f, 0.0f);
  if (q8) oqp = load_qparam(d.oq);
  float sc_[WN], bias_[WN];
  int corr_[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = n0 + (wn * WN + j) * 32 + (lane & 31);
    sc_[j] = 1.0f;
    bias_[j] = 0.0f;
    corr_[j] = 0;
    if (n < d.Cout) {
      const int4 wmv = reinterpret_cast<const int4*>(d.wmeta)[n];
      corr_[j] = (128 - za) * (wmv.y - p.Ktot * wmv.x);
      sc_[j] = aqp.x * d.wscale[n];
      bias_[j] = d.bias ? d.bias[n] : 0.0f;
    }
  }
  const int gq = tid / TPR, c4 = (tid % TPR) * 4;      // phase 2: wave-row group, first of 4 channels (tid < 4 * TPR)
  // phase 1 of pass (i, g): 4 accumulator registers of every N-tile -> LDS.  Instantiated per pass (the registers must be
  // static); everything else of a pass is shared code in a rolled loop -- eight copies of the store pass were ~280 KB
  // of instructions, far beyond the instruction cache
  auto stage = [&](auto pass_tag) {
    constexpr int pass = decltype(pass_tag)::value;
    constexpr int i = pass >> 2, g = pass & 3;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = (wn * WN + j) * 32 + (lane & 31);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        ldsO[(wm * 8 + rr + 4 * h) * LDO + col] = sc_[j] * static_cast<float>(acc[i][j][g * 4 + rr] + corr_[j]) + bias_[j];
    }
  };
#pragma unroll 1
  for (int pass = 0; pass < 8; ++pass) {
    const int i = pass >> 2, g = pass & 3;
    // fp16-stream store pass: its residual rows (and the image's temb row) are requested here, branch-free, so that their
    // latency runs under the staging below instead of forming a chain of 8 dependent waits in the row loop
    constexpr int TPR8 = BN / 8;
    const bool act8 = o16 && h8 && tid < 4 * TPR8;
    const int gq8 = tid / TPR8, c8 = (tid % TPR8) * 8;
    uint4 rres[8];
    float4 ra0 = make_float4(0.f, 0.f, 0.f, 0.f), ra1 = ra0;
    if (act8) {
      const int row0 = gq8 * 64 + i * 32 + g * 8;
      const int nc = (n0 + c8) < d.Cout ? (n0 + c8) : 0;
      if (d.residual && d.res_f16) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int mc = (m0 + row0 + k) < p.M ? (m0 + row0 + k) : p.M - 1;
          rres[k] = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(d.residual) + static_cast<size_t>(mc) * d.Cout + nc);
        }
      }
      if (rowadd) {
        const int mc = (m0 + row0) < p.M ? (m0 + row0) : p.M - 1;      // the 8 rows of a group belong to one image (8 | H*W)
        ra0 = *reinterpret_cast<const float4*>(rowadd + static_cast<size_t>(mc / hw) * d.rowadd_ld + nc);
        ra1 = *reinterpret_cast<const float4*>(rowadd + static_cast<size_t>(mc / hw) * d.rowadd_ld + nc + 4);
      }
    }
    __syncthreads();            // previous pass consumed (first pass: every wave has left the K loop)
    switch (pass) {
      case 0: stage(std::integral_constant<int, 0>{}); break;
      case 1: stage(std::integral_constant<int, 1>{}); break;
      case 2: stage(std::integral_constant<int, 2>{}); break;
      case 3: stage(std::integral_constant<int, 3>{}); break;
      case 4: stage(std::integral_constant<int, 4>{}); break;
      case 5: stage(std::integral_constant<int, 5>{}); break;
      case 6: stage(std::integral_constant<int, 6>{}); break;
      default: stage(std::integral_constant<int, 7>{}); break;
    }
    __syncthreads();
    if (q8 && q16) {
      // int8 output: items of one row x 16 channels (16-byte stores), balanced over the block
      for (int item = tid; item < 32 * (BN / 16); item += 512) {
        const int sr = item / (BN / 16), c16 = (item - sr * (BN / 16)) * 16;
        const int m = m0 + (sr >> 3) * 64 + i * 32 + g * 8 + (sr & 7), n = n0 + c16;
        if (m >= p.M || n >= d.Cout) continue;
        unsigned w[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float4 v = *reinterpret_cast<const float4*>(ldsO + sr * LDO + c16 + 4 * q4);
          if (rowadd) {
            const float4 a = *reinterpret_cast<const float4*>(rowadd + static_cast<size_t>(m / hw) * d.rowadd_ld + n + 4 * q4);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          }
          if (d.residual) {
            const float4 a = load_res4(d, m, n + 4 * q4);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          }
          w[q4] = pack_q4(v.x, v.y, v.z, v.w, oqp);
        }
        *reinterpret_cast<uint4*>(d.yq + static_cast<size_t>(m) * d.Cout + n) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    } else if (o16 && h8) {
      // fp16 activation stream: items of 8 channels (16 bytes out, 16 bytes of fp16 residual in), 8 rows per thread
      if (act8) {
        const int row0 = gq8 * 64 + i * 32 + g * 8, n = n0 + c8;
        float ps[2][8], pss[2][8];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
          for (int q = 0; q < 8; ++q) ps[gi][q] = pss[gi][q] = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int m = m0 + row0 + k;
          if (m >= p.M || n >= d.Cout) continue;
          const float4 v0 = *reinterpret_cast<const float4*>(ldsO + (gq8 * 8 + k) * LDO + c8);
          const float4 v1 = *reinterpret_cast<const float4*>(ldsO + (gq8 * 8 + k) * LDO + c8 + 4);
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          if (rowadd) {
            v[0] += ra0.x; v[1] += ra0.y; v[2] += ra0.z; v[3] += ra0.w; v[4] += ra1.x; v[5] += ra1.y; v[6] += ra1.z; v[7] += ra1.w;
          }
          if (d.residual) {
            if (d.res_f16) {
              const unsigned uw[4] = {rres[k].x, rres[k].y, rres[k].z, rres[k].w};
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float2 f = __half22float2(*reinterpret_cast<const __half2*>(&uw[q]));
                v[2 * q] += f.x;
                v[2 * q + 1] += f.y;
              }
            } else {
              const float4 a0 = load_res4(d, m, n), a1 = load_res4(d, m, n + 4);
              v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
            }
          }
          *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(d.y) + static_cast<size_t>(m) * d.ldy + d.y_coff + n) =
              make_uint4(pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7]));
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            ps[k >> 2][q] += v[q];
            pss[k >> 2][q] += v[q] * v[q];
          }
        }
        if (seg) {
          float2* pp = ldsP + (gq8 * 8 + i * 4 + g) * BN + c8;
#pragma unroll
          for (int q = 0; q < 8; ++q) pp[q] = make_float2(ps[0][q] + ps[1][q], pss[0][q] + pss[1][q]);
        }
      }
    } else     if (tid < 4 * TPR) {
      const int row0 = gq * 64 + i * 32 + g * 8, n = n0 + c4;
      float4 ps[2], pss[2];
      ps[0] = ps[1] = pss[0] = pss[1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int m = m0 + row0 + k;
        if (m >= p.M || n >= d.Cout) continue;
        float4 v = *reinterpret_cast<const float4*>(ldsO + (gq * 8 + k) * LDO + c4);
        if (rowadd) {
          const float4 a = *reinterpret_cast<const float4*>(rowadd + static_cast<size_t>(m / hw) * d.rowadd_ld + n);
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        if (d.residual) {
          float4 a;
          if (d.res_f16) {
            const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(d.residual) + static_cast<size_t>(m) * d.Cout + n);
            const float2 lo = __half22float2(*reinterpret_cast<const __half2*>(&u.x)), hi = __half22float2(*reinterpret_cast<const __half2*>(&u.y));
            a = make_float4(lo.x, lo.y, hi.x, hi.y);
          } else {
            a = *reinterpret_cast<const float4*>(d.residual + static_cast<size_t>(m) * d.Cout + n);
          }
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        if (q8) {
          char4 q;
          q = quant_char4(v.x, v.y, v.z, v.w, make_quantp(oqp));
          *reinterpret_cast<char4*>(d.yq + static_cast<size_t>(m) * d.Cout + n) = q;
        } else if (o16) {
          const __half2 lo = __floats2half2_rn(v.x, v.y), hi = __floats2half2_rn(v.z, v.w);
          uint2 u;
          u.x = *reinterpret_cast<const unsigned*>(&lo);
          u.y = *reinterpret_cast<const unsigned*>(&hi);
          *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(d.y) + static_cast<size_t>(m) * d.ldy + d.y_coff + n) = u;
        } else {
          *reinterpret_cast<float4*>(d.y + static_cast<size_t>(m) * d.ldy + d.y_coff + n) = v;
        }
        ps[k >> 2].x += v.x; ps[k >> 2].y += v.y; ps[k >> 2].z += v.z; ps[k >> 2].w += v.w;
        pss[k >> 2].x += v.x * v.x; pss[k >> 2].y += v.y * v.y; pss[k >> 2].z += v.z * v.z; pss[k >> 2].w += v.w * v.w;
      }
      if (seg) {
        // an 8-row group = (rows 0..3 added in order) + (rows 4..7 added in order): the order of every tile shape
        float2* pp = ldsP + (gq * 8 + i * 4 + g) * BN + c4;
        pp[0] = make_float2(ps[0].x + ps[1].x, pss[0].x + pss[1].x);
        pp[1] = make_float2(ps[0].y + ps[1].y, pss[0].y + pss[1].y);
        pp[2] = make_float2(ps[0].z + ps[1].z, pss[0].z + pss[1].z);
        pp[3] = make_float2(ps[0].w + ps[1].w, pss[0].w + pss[1].w);
      }
    }
  }
  if (seg) {
    __syncthreads();
    const int nseg = BM / seg, gps = seg / 8;
    for (int o = tid; o < nseg * BN; o += 512) {
      const int sidx = o / BN, col = o - sidx * BN;
      float2 a = make_float2(0.0f, 0.0f);
      for (int q = 0; q < gps; ++q) {            // a segment = its 8-row groups added in row order
        const float2 b = ldsP[(sidx * gps + q) * BN + col];
        a.x += b.x;
        a.y += b.y;
      }
      const int row0 = m0 + sidx * seg, n = n0 + col;
      if (row0 < p.M && n < d.Cout) reinterpret_cast<float2*>(d.stats)[static_cast<size_t>(row0 / seg) * d.Cout + n] = a;
    }
  }
}

}  // namespace

bool launch_conv_slab(tfmq_handle h, ConvP& p, hipStream_t st, bool forced) {
  const tfmq_conv_desc& d = p.d;
  const int Hv = d.up2x ? 2 * d.H : d.H, Wv = d.up2x ? 2 * d.W : d.W;
  if (d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad_t != 1 || d.pad_l != 1 || d.Ho != Hv || d.Wo != Wv || p.Hv != Hv || p.Wv != Wv) return false;
  if (d.Cin % 64 != 0 || static_cast<size_t>(d.B) * d.H * d.W * d.Cin >= (static_cast<size_t>(1) << 31)) return false;
  if (!(d.out_mode == TFMQ_OUT_F32 || d.out_mode == TFMQ_OUT_Q8 || (d.out_mode == TFMQ_OUT_F16 && !d.yt))) return false;
  if (((d.Cout | d.ldy | d.y_coff) & 3) != 0 || (d.rowadd && (d.rowadd_ld & 3) != 0)) return false;
  if (d.stats && 256 % d.stats_seg != 0) return false;
  SlabP sp;
  sp.HW = Hv * Wv;
  sp.SW = Wv + 2;
  if (sp.HW % 256 == 0 && 256 % Wv == 0) {
    sp.imgs = 1;
    sp.slab_rows = (256 / Wv + 2) * sp.SW;
    sp.SI = sp.slab_rows;
  } else if (256 % sp.HW == 0) {
    sp.imgs = 256 / sp.HW;
    sp.SI = (Hv + 2) * sp.SW;
    sp.slab_rows = sp.imgs * sp.SI;
  } else {
    return false;
  }
  if (sp.slab_rows > SLAB_CAP) return false;
  const int WN = d.Cout % 320 == 0 ? 5 : (d.Cout > 128 ? 4 : 2);
  const int BN = 64 * WN;
  const int tiles_n = (d.Cout + BN - 1) / BN, tiles_m = (p.M + 255) / 256;
  // one 8-wave block per CU: a grid that leaves most CUs idle is better served by the small-tile kernels
  if (!forced && static_cast<long>(tiles_n) * tiles_m < h->cu_count) return false;
  p.tiles_n = tiles_n;
  sp.p = p;
  dim3 grid(static_cast<unsigned>(tiles_n) * tiles_m);
  if (WN == 5) hipLaunchKernelGGL((k_conv3_slab<5>), grid, dim3(512), 0, st, sp);
  else if (WN == 4) hipLaunchKernelGGL((k_conv3_slab<4>), grid, dim3(512), 0, st, sp);
  else hipLaunchKernelGGL((k_conv3_slab<2>), grid, dim3(512), 0, st, sp);
  return true;
}
