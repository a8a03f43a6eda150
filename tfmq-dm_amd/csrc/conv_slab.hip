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
//   * NWM = 2 (round 3): the same pipeline on 128 pixels x (64 * WN) channels and FOUR waves (2 along M x 2 along N), two weight
//     stages instead of three and 320-row slab buffers -> exactly 80 KiB of LDS and <= 256 VGPRs, so that TWO blocks share a CU:
//     one block's register-direct epilogue (a third of a block's life: scratch/phase_slab.py) and its per-K-step barrier stalls run
//     under the other block's MFMAs, and the 8x8 / 16x16 levels get twice as many blocks to spread over 256 CUs.  Measured
//     (scratch/bench_slab.py, UNet batch 128): +8 ... +23 % at the 8x8 level (256 blocks instead of 128), -1 ... -7 % elsewhere
//     (twice the weight DMA per MFMA, two stages) -- the two co-resident blocks run in lockstep, and starting the second one
//     25 / 50 / 75 % of a block time late is monotonically slower (gpurun_out/r03/bench_slab_stagger.txt): a block's life is a chain
//     of its own latencies, not contention with its neighbour.  The engines' per-shape measurement picks the form (ops.py);
//   * both operands travel global -> LDS by LDS-DMA with the 16-byte-slot XOR swizzle of the other kernels (on the
//     source address and on the fragment reads), the slab double-buffered per chunk, the weights in three K-step
//     stages, counted vmcnt and one raw s_barrier per K-step.
// int32 accumulation is exact, so the result equals k_conv_dma's bit for bit whatever the K order; the epilogue
// (dequantise, + bias, + temb row, + residual, GroupNorm statistics in the one summation order every tile shape
// uses) stages 32 tile rows at a time through LDS and writes whole rows.
#include "conv_common.hpp"
#include <type_traits>
#include <cstdlib>
#ifdef TFMQ_PHASE_TIMERS
#include <cstdio>
#include <cstdlib>
#include <vector>
#define SLAB_MARK(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
#else
#define SLAB_MARK(i) do { } while (0)
#endif

// Timing-only ablations of the K loop (scratch/r04_slab_abl.sh builds one library per mask with -DSLAB_ABLATE=<mask>; results are garbage):
// 1 no DMA issue after the first chunk's, 2 no MFMAs, 4 no fragment reads (operands from registers), 8 no per-K-step barrier / waits
#ifndef SLAB_ABLATE
#define SLAB_ABLATE 0
#endif
#ifndef SLAB_PREFETCH
#define SLAB_PREFETCH 1
#endif

namespace {

struct SlabP {
  ConvP p;
  int HW;         // pixels per image
  int imgs;       // images per 256-pixel tile (> 1 when an image has fewer than 256 pixels)
  int SW;         // slab row width = W + 2
  int SI;         // slab rows per image (imgs > 1); imgs == 1: rows of the whole slab
  int slab_rows;  // rows of the slab that carry pixels (<= 512)
  int issue_split; // TFMQ_SLAB_ISSUE_SPLIT=1 (round 6 A/B): the second-dispatched half of the waves issues its LDS-DMA pieces right behind the step's barrier,
                  // the first half behind its first-half MFMAs (as every wave did): the 24-piece burst of a step (~17 cycles of the CU's L1 -> LDS path per piece,
                  // profiles/r06_ubench_ldsdma_stream.txt) was queued in front of every wave at once -- 660 ... 720 cycles of a 2400-cycle step (profiles/r06_kstep_slab.txt)
  int stats_lds;  // GroupNorm statistics through a wave-private LDS transpose (round 6; TFMQ_SLAB_STATS_LDS=0: the DPP sums of round 2)
  int prio;       // TFMQ_SETPRIO=1 (A/B runs): the second-dispatched half of an 8-wave block runs at s_setprio 1 (MI355X_MICROARCH.md: static priority)
};

// Geometry of a variant: NWM waves along M (4: 256-pixel tiles, 8 waves, one block per CU; 2: 128-pixel tiles, 4 waves, two per CU)
// PP (round 6): the two waves of a SIMD in barrier-enforced anti-phase (see the kernel), FOUR weight stages
template <int NWM, int PP = 0>
struct SlabGeo {
  static constexpr int NW = 2 * NWM;                 // waves per block
  static constexpr int NT = 64 * NW;                 // threads
  static constexpr int BM = 64 * NWM;                // pixels per tile
  static constexpr int NST = PP ? 4 : (NWM == 4 ? 3 : 2);       // weight K-step stages in LDS
  static constexpr int NIT = NWM == 4 ? 4 : 5;       // slab DMA pieces per wave and chunk (one per K-step of taps 0 .. NIT-1)
  static constexpr int CAP = NIT * NW * 16;          // rows per slab buffer: 512 / 320
  static constexpr int STRIDE = CAP * 64;            // 32 KiB / 20 KiB
};

// Statistics through LDS (round 6).  The DPP form (group8_sum) spends 72 VALU instructions per register octet -- four dependent DPP adds per
// value and statistic, of which only the lanes with lane % 8 == 0 keep the result: 55 % of the epilogue's VALU work, and the epilogue is a third
// of a short-K block's life (profiles/r06_phase_slab.txt).  Here a wave writes the fp32 values of a 32-row x 32-channel tile to a private block
// (rows 144 bytes apart: conflict-free 16-byte writes and 4-byte column reads) and every lane sums ONE (8-row group, channel) column for two
// groups, serially, in the canonical order ((r0 + r1) + r2) + r3 + (((r4 + r5) + r6) + r7) -- the same additions on the same values, so the
// partial sums are bit-identical (tests/test_conv_epilogue_modes_gpu.py compares every tile kernel's statistics): 44 VALU + 20 DS per tile
// instead of 144 VALU.
constexpr int SLAB_T_ROW = 36;                       // dwords per transposed row: 32 channels + 4
constexpr int SLAB_T_BYTES = 32 * SLAB_T_ROW * 4;    // per wave

template <int WN, int NWM, int PP = 0>
__host__ __device__ constexpr int slab_lds_bytes() {
  using G = SlabGeo<NWM, PP>;
  constexpr int BN = 64 * WN;
  constexpr int main_ = 2 * G::STRIDE + G::NST * BN * 64;
  // 8-row-group partial sums + the table of per-column constants + a 32-row x 32-channel fp32 transpose block per wave (statistics)
  constexpr int epi = (G::BM / 8) * BN * 8 + (3 + 4) * BN * 4 + G::NW * SLAB_T_BYTES;
  return main_ > epi ? main_ : epi;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// F16OP: the un-quantised / weight-only 3x3 layers on the same pipeline -- fp16 activations (a 64-byte slab row = 32 channels), fp16
// weights [cout][tap][cin_pad] row-major (tfmq_pack_w_f16), v_mfma_f32_32x32x16_f16, value = scale * acc + bias (k_conv_dma<true>'s
// arithmetic; the K order differs from its tap-major one, so the two agree to fp32 summation noise, not bit for bit).
//
// PP = true (round 6, 8-wave form only): PING-PONG.  In the one-barrier-per-step loop above the two waves of a SIMD (w and w + 4) run in
// lockstep: both read fragments, both issue their 20 MFMAs, both issue LDS-DMA -- and an in-order wave issues no MFMA while it does anything
// else, so the matrix pipe idles through every wave's reads / DMA issue / waits (profiles/r06_kstep_slab.txt: 1280 of ~2400 cycles of a
// K-step are MFMA time on a SIMD).  Here waves 0-3 and waves 4-7 run the same step HALF A STEP APART, two barriers per step:
//     load phase     fragment reads of step s, LDS-DMA issue for step s + 3, counted wait for the pieces of step s + 1
//     compute phase  the 20 MFMAs of step s, nothing else
// so on every SIMD one wave's compute phase lies beside the other's load phase (the "compute | load" pairing of MI355X_MICROARCH.md,
// two waves per SIMD).  Visibility: a wave's pieces for step s are issued in its load phase of step s - 3, waited for at the END of its
// load phase of step s - 1 (in-order vmcnt: everything but what it issued in its last two load phases may still fly) and a barrier closes
// that phase -- before the first fragment read of step s by either group.  Four weight stages (stage = step & 3): the stage a load phase
// fills was last read two phases earlier.  Same MFMAs on the same operands in the same order per accumulator: bit-identical.
template <int WN, bool F16OP = false, int NWM = 4, int PP = 0>
__global__ __launch_bounds__(64 * 2 * NWM, 2) void k_conv3_slab(SlabP sp) {
  static_assert(!PP || NWM == 4, "ping-pong needs two waves per SIMD in one block");
  using G = SlabGeo<NWM, PP>;
  constexpr int NW = G::NW, NT = G::NT, NST = G::NST, NIT = G::NIT, SLAB_BYTES = G::STRIDE;
  constexpr int BM = G::BM, BN = 64 * WN;
  constexpr int BST = BN * 64;                 // bytes of one weight K-step stage
  constexpr int BOFF = 2 * SLAB_BYTES;
  constexpr int NBP = BN / 16;                 // 1-KiB weight pieces per K-step
  constexpr int MAXCH = (NBP + NW - 1) / NW;   // weight pieces a wave moves per K-step
  __shared__ __attribute__((aligned(1024))) unsigned char lds[slab_lds_bytes<WN, NWM, PP>()];

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

  SLAB_MARK(0);
  float2 aqp = make_float2(1.0f, 128.0f);
  if constexpr (!F16OP) aqp = load_qparam(d.aq);
  constexpr int EB = F16OP ? 2 : 1;              // bytes per activation element

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
  // slab DMA sources: piece it*NW + wid, 16 rows each, this lane's row = piece*16 + lane/4
  const unsigned char* xb = static_cast<const unsigned char*>(d.x);
  const int dcol = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;        // swizzle on the SOURCE: (row >> 2) & 3 == (lane >> 4) & 3
  int s_off[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int j = (it * NW + wid) * 16 + (lane >> 2);
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
  const unsigned char* padp = p.pad_table + (F16OP ? 0u : (static_cast<unsigned>(za - 128) & 0xffu)) * 64 + dcol;   // fp16: the zero row
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));

  auto issue_slab = [&](int it, int c, int buf) {
    int so = s_off[it];
    // (PP: the offset is made opaque at every use -- the compiler otherwise hoists four 64-bit per-lane source pointers out of the K loop,
    // spills them, and the reload inside the loop is a scratch load behind an s_waitcnt vmcnt(0): a full drain of the LDS-DMA queue per chunk)
    if constexpr (PP) asm volatile("" : "+v"(so));
    const unsigned char* src = so >= 0 ? xb + static_cast<size_t>(static_cast<unsigned>(so)) * EB + c * 64 + dcol : padp;
    glds16(src, lds0 + buf * SLAB_BYTES + __builtin_amdgcn_readfirstlane((it * NW + wid) * 1024));
  };

  // weight DMA sources: pieces wid, wid + NW, wid + 2 NW ... (< NBP): per-lane 32-bit offsets from the (uniform) weight base
  const unsigned char* wbase = static_cast<const unsigned char*>(d.w);
  unsigned b_off[MAXCH];
#pragma unroll
  for (int k = 0; k < MAXCH; ++k) {
    const int piece = wid + NW * k;
    int n = n0 + piece * 16 + (lane >> 2);
    if constexpr (F16OP) {
      n = n < d.Cout ? n : d.Cout - 1;
      b_off[k] = static_cast<unsigned>(static_cast<size_t>(n) * 9 * p.cin_pad * 2 + dcol);
    } else {
      n = n < p.cout_pad ? n : p.cout_pad - 1;
      b_off[k] = static_cast<unsigned>((static_cast<size_t>(n / 32) * p.nsteps * 32 + (n % 32)) * 64 + dcol);
    }
  }

  // fragment read offsets
  // weight rows in the permuted order of lin_brow: lane half h then owns 16 consecutive channels of every 32-channel tile
#ifdef TFMQ_DBG_NATURAL_ROWS      // timing experiment only (wrong channel order): are the permuted fragment rows slower to read?
  const int brow = lane & 31;
#else
  const int brow = lin_brow(lane & 31);
#endif
  const int b_rel = (wn * WN * 32 + brow) * 64 + ((h ^ ((brow >> 2) & 3)) << 4);
  int slab_toggle = 0;

  v16i acc[2][WN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

#ifdef TFMQ_PHASE_TIMERS
  // where a wave's K-step goes (shader cycles, wave 0): [0] counted vmcnt wait, [1] barrier, [2] fragment reads until their data is there,
  // [3] first half's MFMA issue, [4] DMA issue of the next steps, [5] second half's MFMA issue, [6] steps
  unsigned long long kacc[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long kt = clock64();
#define SKT(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = clock64(); kacc[i] += t_ - kt; kt = t_; } while (0)
#else
#define SKT(i) do { } while (0)
#endif
  auto kloop = [&](auto bch_tag) {
    constexpr int B_CH = decltype(bch_tag)::value;
    auto issue_b = [&](int c, int tap, int stage) {
      const size_t boff = F16OP ? static_cast<size_t>(tap * p.cin_pad + c * 32) * 2 : static_cast<size_t>(tap * p.chunks + c) * 2048;
#pragma unroll
      for (int k = 0; k < B_CH; ++k)
        glds16_sv(wbase + boff, b_off[k], lds0 + BOFF + stage * BST + __builtin_amdgcn_readfirstlane((wid + NW * k) * 1024));
    };
    auto step = [&](int c, auto tap_tag) {
      constexpr int TAP = decltype(tap_tag)::value;
      // Loads younger than this step's weights: the weights of the NST - 2 steps after it and the slab pieces issued in the
      // NST - 1 steps before this one (taps 0 .. NIT-1 carry one each)
      constexpr int X = ((TAP >= 1 && TAP <= NIT) ? 1 : 0) + ((NST == 3 && TAP >= 2 && TAP <= NIT + 1) ? 1 : 0);
      const int pos = c * 9 + TAP;
      if constexpr (!(SLAB_ABLATE & 8)) {
        if constexpr (SLAB_ABLATE & 1) wait_vmcnt<0>();
        else if (pos + 1 < p.nsteps) wait_vmcnt<(NST - 2) * B_CH + X>();
        else wait_vmcnt<0>();
        SKT(0);
        asm volatile("s_barrier" ::: "memory");
        SKT(1);
      }
      // stage of a K-step: NST = 3 -> tap % 3 (9 taps a chunk); NST = 2 -> parity of the step index
      const int st_next = NST == 3 ? (TAP + 2) % 3 : ((pos + 1) & 1);
      auto issue_next = [&]() {
        if constexpr (SLAB_ABLATE & 1) return;
        if (pos + NST - 1 < p.nsteps) {
          if constexpr (TAP + NST - 1 < 9) issue_b(c, TAP + NST - 1, st_next);
          else issue_b(c + 1, TAP + NST - 1 - 9, st_next);
        }
        if constexpr (TAP < NIT) {
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
        const int ar = (srow << 6) + (((h ^ (srow >> 2)) & 3) << 4);
        a_rel[i] = (SLAB_BYTES & (SLAB_BYTES - 1)) == 0 ? (ar ^ slab_toggle) : (ar + slab_toggle);     // 32 KiB buffers toggle by XOR
      }
      const unsigned char* sb = lds + BOFF + (NST == 3 ? TAP % 3 : (pos & 1)) * BST;
      // Both halves' fragments are requested up front (SLAB_PREFETCH, round 4): the ks = 1 reads then travel under the ks = 0 MFMAs instead
      // of standing between the two halves (timing ablation: the fragment reads were 16 ... 28 % of the launch, scratch/r04_slab_abl.sh).
      // 28 more live registers inside the loop, which the epilogue's budget of 256 covers.
      v4i af[2][2], bf[2][WN];
      auto read_frags = [&](int ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if constexpr (SLAB_ABLATE & 4) af[ks][i] = v4i{a_rel[i], ks, i, lane};
          else af[ks][i] = *reinterpret_cast<const v4i*>(lds + (a_rel[i] ^ (ks << 5)));
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          if constexpr (SLAB_ABLATE & 4) bf[ks][j] = v4i{b_rel, ks, j, lane};
          else bf[ks][j] = *reinterpret_cast<const v4i*>(sb + ((b_rel ^ (ks << 5)) + j * 2048));
        }
      };
      const bool early = sp.issue_split && wid >= NW / 2;      // (wave-uniform)
      if (early) issue_next();
      read_frags(0);
      if constexpr (SLAB_PREFETCH) read_frags(1);
#ifdef TFMQ_PHASE_TIMERS
      asm volatile("" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(bf[0][0]), "+v"(bf[0][WN - 1]), "+v"(bf[1][WN - 1]));
      SKT(2);
#endif
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (!SLAB_PREFETCH) {
          if (ks == 1) read_frags(1);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            if constexpr (F16OP) {
              typedef _Float16 v8h_t __attribute__((ext_vector_type(8)));
              typedef float v16f_t __attribute__((ext_vector_type(16)));
              v16f_t& af32 = *reinterpret_cast<v16f_t*>(&acc[i][j]);
              af32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<v8h_t*>(&bf[ks][j]), *reinterpret_cast<v8h_t*>(&af[ks][i]), af32, 0, 0, 0);
            } else if constexpr (SLAB_ABLATE & 2) {
              acc[i][j][(ks * 2 + i) & 15] += bf[ks][j][0] ^ af[ks][i][1];
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);      // (channels x pixels): lane = pixel
            }
          }
        // the DMA issue of the next K-steps (SALU M0 moves + VMEM, ~100 clk a piece) sits behind the first half's MFMAs:
        // the matrix pipe works through them while the wave issues the loads, instead of idling right after the barrier
#ifdef TFMQ_PHASE_TIMERS
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][WN - 1]));
        if (ks == 0) SKT(3);
        else { SKT(5); kacc[6] += 1; }
#endif
        if (ks == 0) {
          if (!early) issue_next();
          SKT(4);
        }
      }
    };
#pragma unroll
    for (int it = 0; it < NIT; ++it) issue_slab(it, 0, 0);
    issue_b(0, 0, 0);
    if constexpr (NST == 3) issue_b(0, 1, 1);
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
      slab_toggle = SLAB_BYTES - slab_toggle;
    }
  };
  // ---- the ping-pong loop (PP)
  auto kloop_pp = [&](auto bch_tag) {
    constexpr int B_CH = decltype(bch_tag)::value;
    const int grp = wid >> 2;                          // waves w and w + 4 share a SIMD (wave-uniform)
    auto issue_b = [&](int c, int tap, int stage) {
      const size_t boff = F16OP ? static_cast<size_t>(tap * p.cin_pad + c * 32) * 2 : static_cast<size_t>(tap * p.chunks + c) * 2048;
#pragma unroll
      for (int k = 0; k < B_CH; ++k)
        glds16_sv(wbase + boff, b_off[k], lds0 + BOFF + stage * BST + __builtin_amdgcn_readfirstlane((wid + NW * k) * 1024));
    };
    int pos = 0;
    auto step = [&](int c, auto tap_tag) {
      constexpr int TAP = decltype(tap_tag)::value;
      // ---------------- load phase
      constexpr int KH = TAP / 3, KW = TAP % 3;
      const int toff = KH * SW + KW;
      int a_rel[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int srow = srow0[i] + toff;
        const int ar = (srow << 6) + (((h ^ (srow >> 2)) & 3) << 4);
        a_rel[i] = ar ^ slab_toggle;                   // 32 KiB buffers toggle by XOR
      }
      const unsigned char* sb = lds + BOFF + (pos & 3) * BST;
      v4i af[2][2], bf[2][WN];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) af[ks][i] = *reinterpret_cast<const v4i*>(lds + (a_rel[i] ^ (ks << 5)));
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[ks][j] = *reinterpret_cast<const v4i*>(sb + ((b_rel ^ (ks << 5)) + j * 2048));
      }
      const bool more = pos + 3 < p.nsteps;
      // PP == 1: this wave's LDS-DMA pieces (weights of step s + 3, a slab piece at taps 0 .. NIT-1) are issued HERE, in the load phase.
      // PP == 2: they are issued between the LAST MFMAs of the compute phase (profiles/r06_pp_phases.txt: with the issue in the load phase
      // that phase takes ~1000 cycles beside a 680-cycle compute phase, whose wave then waits ~300 cycles at the barrier).
      auto issue_piece = [&](int k) {        // piece k of this step's batch: weights 0 .. B_CH-1, then the slab piece
        if (k < B_CH) {
          if (more) {
            const int cc = TAP + 3 < 9 ? c : c + 1, tt = TAP + 3 < 9 ? TAP + 3 : TAP + 3 - 9;
            const size_t boff = F16OP ? static_cast<size_t>(tt * p.cin_pad + cc * 32) * 2 : static_cast<size_t>(tt * p.chunks + cc) * 2048;
            glds16_sv(wbase + boff, b_off[k], lds0 + BOFF + ((pos + 3) & 3) * BST + __builtin_amdgcn_readfirstlane((wid + NW * k) * 1024));
          }
        } else {
          if constexpr (TAP < NIT) issue_slab(TAP, c + 1 < p.chunks ? c + 1 : 0, ((c + 1) & 1));
        }
      };
      if constexpr (PP == 1) {
#pragma unroll
        for (int k = 0; k <= B_CH; ++k) issue_piece(k);
      }
      // PP == 1 -- may still fly: what this phase issued and what the previous load phase issued (weights of steps s + 3, s + 2; a slab
      // piece each at taps 0 .. NIT-1).  Everything older -- the weights of step s + 1 among it -- has landed.  The last steps simply drain.
      // PP == 2 -- may still fly: the batch of the previous compute phase (weights of step s + 2, its slab piece) and the slab piece of the
      // one before (issued behind the weights of step s + 1, which must have landed).
      constexpr int X2 = PP == 1 ? (TAP < NIT ? 1 : 0) + ((TAP >= 1 && TAP <= NIT) ? 1 : 0)
                                 : (((TAP + 8) % 9) < NIT ? 1 : 0) + (((TAP + 7) % 9) < NIT ? 1 : 0);
      constexpr int NB2 = PP == 1 ? 2 * B_CH : B_CH;
      const bool counted = PP == 1 ? more : (pos + 2 < p.nsteps && pos >= 2);       // (PP == 2: the first two steps wait for the prologue's batches)
#ifdef TFMQ_PHASE_TIMERS
      // where a wave's step goes (wave 0 = early half, wave 4 = late half): [0] fragment reads + DMA issue until the fragments are there,
      // [1] counted vmcnt wait, [2] barrier closing the load phase, [3] MFMA issue, [4] barrier closing the compute phase, [6] steps
      SKT(0);
      if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB2 + X2) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SKT(1);
      asm volatile("s_barrier" ::: "memory");
      SKT(2);
#else
      if (counted) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NB2 + X2) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- compute phase
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            if constexpr (PP == 2) {
              // one piece in front of each of the last MFMAs but two: the piece's issue (SALU M0 moves + the TA's ~26 cycles) runs under the
              // 32 matrix-pipe cycles of the MFMA issued before it
              constexpr int NP = B_CH + 1;
              const int q = (ks * 2 + i) * WN + j, first = 4 * WN - 2 * NP;
              if (q >= first && ((q - first) & 1) == 0) {
                __builtin_amdgcn_sched_barrier(0);
                issue_piece((q - first) >> 1);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            if constexpr (F16OP) {
              typedef _Float16 v8h_t __attribute__((ext_vector_type(8)));
              typedef float v16f_t __attribute__((ext_vector_type(16)));
              v16f_t& af32 = *reinterpret_cast<v16f_t*>(&acc[i][j]);
              af32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<v8h_t*>(&bf[ks][j]), *reinterpret_cast<v8h_t*>(&af[ks][i]), af32, 0, 0, 0);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
            }
          }
      __builtin_amdgcn_sched_barrier(0);
      ++pos;
#ifdef TFMQ_PHASE_TIMERS
      asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][WN - 1]));
      SKT(3);
#endif
      if (!(grp == 1 && pos == p.nsteps)) asm volatile("s_barrier" ::: "memory");      // (the late group's last compute phase has no partner)
#ifdef TFMQ_PHASE_TIMERS
      SKT(4);
      kacc[6] += 1;
#endif
      __builtin_amdgcn_sched_barrier(0);
    };
    // prologue: chunk 0's slab, the weights of steps 0, 1, 2; the slab and step 0 have landed behind the first barrier
#pragma unroll
    for (int it = 0; it < NIT; ++it) issue_slab(it, 0, 0);
    issue_b(0, 0, 0);
    if (p.nsteps > 1) issue_b(0, 1, 1);
    if (p.nsteps > 2) issue_b(0, 2, 2);
    if (p.nsteps > 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * B_CH) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (grp == 1) asm volatile("s_barrier" ::: "memory");          // half a step behind
    __builtin_amdgcn_sched_barrier(0);
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
      slab_toggle = SLAB_BYTES - slab_toggle;
    }
  };
  if (sp.prio && wid >= NW / 2) __builtin_amdgcn_s_setprio(1);
  if constexpr (PP) {
    if (NBP % NW != 0 && wid < NBP % NW) kloop_pp(std::integral_constant<int, NBP / NW + 1>{});
    else kloop_pp(std::integral_constant<int, NBP / NW>{});
  } else {
    if (NBP % NW != 0 && wid < NBP % NW) kloop(std::integral_constant<int, NBP / NW + 1>{});
    else kloop(std::integral_constant<int, NBP / NW>{});
  }
  SLAB_MARK(1);
#ifdef TFMQ_PHASE_TIMERS
  if (p.dbg2 && tid == 0)
    for (int i = 0; i < 7; ++i) p.dbg2[static_cast<size_t>(blockIdx.x) * 16 + i] = kacc[i];
  if (PP && p.dbg2 && tid == 256)        // the late half's wave 4: rows behind the 65536 block rows
    for (int i = 0; i < 7; ++i) p.dbg2[static_cast<size_t>(65536 + blockIdx.x) * 16 + i] = kacc[i];
  unsigned long long et0 = clock64();
#endif

  // ================================================================================ epilogue (out of the registers)
  // acc[i][j][8u .. 8u+7] = channels (wn*WN + j)*32 + 16h + 8u .. +7 of pixel row wm*64 + i*32 + lane%32 (lin_brow).
  // No LDS staging and no barriers between the K loop and the stores (the staged epilogue -- eight passes of stage /
  // barrier / 160 of 512 threads storing -- took 34 us of a 133 us block without a residual and 75 us of 121 us with one:
  // scratch/phase_slab.py): every lane converts its own values, 16-byte stores and residual loads, the fp16 residual
  // octets requested three channel tiles ahead, statistics by DPP sums over the 8 lanes of a pixel-row group in the
  // canonical order.
  float2* ldsP = reinterpret_cast<float2*>(lds);                         // [BM / 8 eight-row groups][BN] (sum, sum of squares)
  float* cs = reinterpret_cast<float*>(lds + (BM / 8) * BN * 8);         // scale[BN], corr[BN] (int), bias[BN], rowadd[imgs][BN]
  const int hw = sp.HW;
  const float* rowadd = d.rowadd;
  if (rowadd && d.rowadd_step) rowadd += static_cast<size_t>(load_scalar_i32(d.rowadd_step)) * d.rowadd_step_stride;
  const int seg = d.stats ? d.stats_seg : 0;
  const bool q8 = d.out_mode == TFMQ_OUT_Q8, o16 = d.out_mode == TFMQ_OUT_F16;
  float2 oqp = make_float2(1.0f, 0.0f);
  if (q8) oqp = load_qparam(d.oq);
  const QuantP qP = make_quantp(oqp);
  float* Tw = reinterpret_cast<float*>(lds + (BM / 8) * BN * 8 + (3 + 4) * BN * 4 + wid * SLAB_T_BYTES);      // this wave's transpose block
  const bool st_lds = seg != 0 && sp.stats_lds != 0;
  __syncthreads();                       // every wave has left the K loop: its LDS becomes the table and the partial sums
  for (int tc = tid; tc < BN; tc += NT) {
    const int n = n0 + tc;
    float c_sc = 1.0f, c_bias = 0.0f;
    int c_corr = 0;
    if (n < d.Cout) {
      if constexpr (F16OP) {
        c_sc = d.wscale ? d.wscale[n] : 1.0f;
      } else {
        const int4 wmv = reinterpret_cast<const int4*>(d.wmeta)[n];
        c_corr = (128 - za) * (wmv.y - p.Ktot * wmv.x);
        c_sc = aqp.x * d.wscale[n];
      }
      c_bias = d.bias ? d.bias[n] : 0.0f;
    }
    cs[tc] = c_sc;
    reinterpret_cast<int*>(cs)[BN + tc] = c_corr;
    cs[2 * BN + tc] = c_bias;
    if (rowadd) {
      for (int k = 0; k < sp.imgs; ++k) {
        const int b = (b0 + k) < d.B ? (b0 + k) : d.B - 1;
        cs[(3 + k) * BN + tc] = n < d.Cout ? rowadd[static_cast<size_t>(b) * d.rowadd_ld + n] : 0.0f;
      }
    }
  }
  const bool res16 = d.residual && d.res_f16;
  constexpr int NS = 2 * WN, PD = 3;       // (i, j) steps of a wave; prefetch distance of the fp16 residual octets
  uint4 rq[PD][2];
  auto rrow = [&](int i) { const int m = m0 + wm * 64 + i * 32 + (lane & 31); return m < p.M ? m : p.M - 1; };
  auto load_rq = [&](int st, uint4 (&dst)[2]) {
    const int i = st / WN, j = st - i * WN;
    const size_t base = static_cast<size_t>(rrow(i)) * d.Cout;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int n = n0 + (wn * WN + j) * 32 + 16 * h + 8 * u;
      dst[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(d.residual) + base + (n < d.Cout ? n : 0));
    }
  };
  if (res16) {
#pragma unroll
    for (int st = 0; st < PD; ++st) load_rq(st, rq[st]);
  }
  __syncthreads();                       // table visible
#ifdef TFMQ_PHASE_TIMERS
  unsigned long long et1 = clock64();
#endif

#pragma unroll
  for (int st = 0; st < NS; ++st) {
    const int i = st / WN, j = st - i * WN;
    const int ml = wm * 64 + i * 32 + (lane & 31), m = m0 + ml;
    const bool mok = m < p.M;
    const int im = sp.imgs == 1 ? 0 : ml / hw;
    uint4 rcur[2] = {rq[st % PD][0], rq[st % PD][1]};
    if (res16 && st + PD < NS) load_rq(st + PD, rq[st % PD]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ct = (wn * WN + j) * 32 + 16 * h + 8 * u, n = n0 + ct;
      const bool ok = mok && n < d.Cout;
      f2 v[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float4 sc = *reinterpret_cast<const float4*>(cs + ct + 4 * e);
        const int4 kc = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(cs) + BN + ct + 4 * e);
        const float4 bb = *reinterpret_cast<const float4*>(cs + 2 * BN + ct + 4 * e);
        const v16i& a = acc[i][j];
        if constexpr (F16OP) {
          v[2 * e] = f2{sc.x, sc.y} * f2{__int_as_float(a[8 * u + 4 * e]), __int_as_float(a[8 * u + 4 * e + 1])} + f2{bb.x, bb.y};
          v[2 * e + 1] = f2{sc.z, sc.w} * f2{__int_as_float(a[8 * u + 4 * e + 2]), __int_as_float(a[8 * u + 4 * e + 3])} + f2{bb.z, bb.w};
        } else {
          v[2 * e] = f2{sc.x, sc.y} * f2{static_cast<float>(a[8 * u + 4 * e] + kc.x), static_cast<float>(a[8 * u + 4 * e + 1] + kc.y)} + f2{bb.x, bb.y};
          v[2 * e + 1] = f2{sc.z, sc.w} * f2{static_cast<float>(a[8 * u + 4 * e + 2] + kc.z), static_cast<float>(a[8 * u + 4 * e + 3] + kc.w)} + f2{bb.z, bb.w};
        }
        if (rowadd) {
          const float4 ra = *reinterpret_cast<const float4*>(cs + (3 + im) * BN + ct + 4 * e);
          v[2 * e] += f2{ra.x, ra.y};
          v[2 * e + 1] += f2{ra.z, ra.w};
        }
      }
      if (d.residual) {
        if (d.res_f16) {
          const unsigned uw[4] = {rcur[u].x, rcur[u].y, rcur[u].z, rcur[u].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float2 f = __half22float2(*reinterpret_cast<const __half2*>(&uw[q]));
            v[q] += f2{f.x, f.y};
          }
        } else if (ok) {
          const float4 a0 = *reinterpret_cast<const float4*>(d.residual + static_cast<size_t>(m) * d.Cout + n);
          const float4 a1 = *reinterpret_cast<const float4*>(d.residual + static_cast<size_t>(m) * d.Cout + n + 4);
          v[0] += f2{a0.x, a0.y};
          v[1] += f2{a0.z, a0.w};
          v[2] += f2{a1.x, a1.y};
          v[3] += f2{a1.z, a1.w};
        }
      }
      if (ok) {
        if (q8) {
          *reinterpret_cast<uint2*>(d.yq + static_cast<size_t>(m) * d.Cout + n) =
              make_uint2(quant_pack4(v[0].x, v[0].y, v[1].x, v[1].y, qP), quant_pack4(v[2].x, v[2].y, v[3].x, v[3].y, qP));
        } else if (o16) {
          *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(d.y) + static_cast<size_t>(m) * d.ldy + d.y_coff + n) =
              make_uint4(pack_h2(v[0].x, v[0].y), pack_h2(v[1].x, v[1].y), pack_h2(v[2].x, v[2].y), pack_h2(v[3].x, v[3].y));
        } else {
          float* dst = d.y + static_cast<size_t>(m) * d.ldy + d.y_coff + n;
          *reinterpret_cast<float4*>(dst) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
        }
      }
      if (st_lds) {
        float* dst = Tw + (lane & 31) * SLAB_T_ROW + 16 * h + 8 * u;
        *reinterpret_cast<float4*>(dst) = ok ? make_float4(v[0].x, v[0].y, v[1].x, v[1].y) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *reinterpret_cast<float4*>(dst + 4) = ok ? make_float4(v[2].x, v[2].y, v[3].x, v[3].y) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      } else if (seg) {
        // per channel: the 8-row group's (sum, sum of squares) of the fp32 values (before any rounding of the output);
        // rows / columns outside the tensor add exact zeros
        const int grp = wm * 8 + i * 4 + ((lane & 31) >> 3);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = ok ? v[e].x : 0.0f, x1 = ok ? v[e].y : 0.0f;
          const float s0 = group8_sum(x0), s1 = group8_sum(x1);
          const float q0 = group8_sum(x0 * x0), q1 = group8_sum(x1 * x1);
          if ((lane & 7) == 0) *reinterpret_cast<float4*>(ldsP + grp * BN + ct + 2 * e) = make_float4(s0, q0, s1, q1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (st_lds) {
      // column sums of the tile just written: lane = (channel c, group pair), two 8-row groups each
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      const int c = lane & 31;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int g = 2 * h + k;
        const float* col = Tw + (g * 8) * SLAB_T_ROW + c;
        const float x0 = col[0], x1 = col[SLAB_T_ROW], x2 = col[2 * SLAB_T_ROW], x3 = col[3 * SLAB_T_ROW];
        const float x4 = col[4 * SLAB_T_ROW], x5 = col[5 * SLAB_T_ROW], x6 = col[6 * SLAB_T_ROW], x7 = col[7 * SLAB_T_ROW];
        const float sa = ((x0 + x1) + x2) + x3, sb = ((x4 + x5) + x6) + x7;
        const float qa = ((x0 * x0 + x1 * x1) + x2 * x2) + x3 * x3, qb = ((x4 * x4 + x5 * x5) + x6 * x6) + x7 * x7;
        ldsP[(wm * 8 + i * 4 + g) * BN + (wn * WN + j) * 32 + c] = make_float2(sa + sb, qa + qb);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#ifdef TFMQ_PHASE_TIMERS
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long et2 = clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long et3 = clock64();
  if (p.dbg2 && tid == 0) {
    p.dbg2[static_cast<size_t>(blockIdx.x) * 16 + 8] = et1 - et0;      // constants table + residual prefetch + two barriers
    p.dbg2[static_cast<size_t>(blockIdx.x) * 16 + 9] = et2 - et1;      // the (i, j) loop: affine map, residual, stores issued, statistics
    p.dbg2[static_cast<size_t>(blockIdx.x) * 16 + 10] = et3 - et2;    // until the last store is acknowledged
  }
#endif
  if (seg) {
    __syncthreads();
    const int nseg = BM / seg, gps = seg / 8;
    for (int o = tid; o < nseg * BN; o += NT) {
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
  SLAB_MARK(2);
}

}  // namespace

bool launch_conv_slab(tfmq_handle h, ConvP& p, hipStream_t st, bool forced, bool f16, bool half_m) {
  const tfmq_conv_desc& d = p.d;
  const int Hv = d.up2x ? 2 * d.H : d.H, Wv = d.up2x ? 2 * d.W : d.W;
  if (d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad_t != 1 || d.pad_l != 1 || d.Ho != Hv || d.Wo != Wv || p.Hv != Hv || p.Wv != Wv) return false;
  if (d.Cin % 32 != 0 || (!f16 && p.chunks != (d.Cin + 63) / 64) ||
      static_cast<size_t>(d.B) * d.H * d.W * d.Cin * (f16 ? 2 : 1) >= (static_cast<size_t>(1) << 31)) return false;
  if (f16 && (!d.x_f16 || p.cin_pad != d.Cin || p.chunks != d.Cin / 32 || d.out_mode == TFMQ_OUT_Q8)) return false;
  if (!(d.out_mode == TFMQ_OUT_F32 || d.out_mode == TFMQ_OUT_Q8 || (d.out_mode == TFMQ_OUT_F16 && !d.yt))) return false;
  if (((d.Cout | d.ldy | d.y_coff) & 7) != 0) return false;             // a lane moves whole 8-channel octets
  const int BM = half_m ? 128 : 256, cap = half_m ? SlabGeo<2>::CAP : SlabGeo<4>::CAP;
  if (d.stats && BM % d.stats_seg != 0) return false;
  SlabP sp;
  static const int prio_env = getenv("TFMQ_SETPRIO") ? atoi(getenv("TFMQ_SETPRIO")) : 0;
  sp.prio = prio_env;
  static const int issue_split_env = getenv("TFMQ_SLAB_ISSUE_SPLIT") ? atoi(getenv("TFMQ_SLAB_ISSUE_SPLIT")) : 0;
  sp.issue_split = issue_split_env;
  static const int pp_env = getenv("TFMQ_SLAB_PP") ? atoi(getenv("TFMQ_SLAB_PP")) : 1;      // ping-pong K loop of the 8-wave w4a8 form (round 6)
  static const int stats_lds_env = getenv("TFMQ_SLAB_STATS_LDS") ? atoi(getenv("TFMQ_SLAB_STATS_LDS")) : 1;
  sp.stats_lds = stats_lds_env;
  sp.HW = Hv * Wv;
  sp.SW = Wv + 2;
  if (sp.HW % BM == 0 && BM % Wv == 0) {
    sp.imgs = 1;
    sp.slab_rows = (BM / Wv + 2) * sp.SW;
    sp.SI = sp.slab_rows;
  } else if (BM % sp.HW == 0) {
    sp.imgs = BM / sp.HW;
    sp.SI = (Hv + 2) * sp.SW;
    sp.slab_rows = sp.imgs * sp.SI;
  } else {
    return false;
  }
  if (sp.slab_rows > cap) return false;
  const int WN = d.Cout % 320 == 0 ? 5 : (d.Cout > 128 ? 4 : 2);
  const int BN = 64 * WN;
  const int tiles_n = (d.Cout + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  // one 8-wave block per CU: a grid that leaves most CUs idle is better served by the small-tile kernels
  if (!forced && static_cast<long>(tiles_n) * tiles_m < h->cu_count) return false;
  p.tiles_n = tiles_n;
  sp.p = p;
  dim3 grid(static_cast<unsigned>(tiles_n) * tiles_m);
#ifdef TFMQ_PHASE_TIMERS
  static unsigned long long* dbuf = nullptr;
  if (!dbuf) (void)hipMalloc(reinterpret_cast<void**>(&dbuf), sizeof(unsigned long long) * 4 * (1u << 16));
  sp.p.dbg = grid.x <= (1u << 16) ? dbuf : nullptr;
  static unsigned long long* dbuf3 = nullptr;
  if (!dbuf3) (void)hipMalloc(reinterpret_cast<void**>(&dbuf3), sizeof(unsigned long long) * 16 * (2u << 16));
  sp.p.dbg2 = grid.x <= (1u << 16) ? dbuf3 : nullptr;
#endif
  if (half_m) {
    if (f16) {
      if (WN == 5) hipLaunchKernelGGL((k_conv3_slab<5, true, 2>), grid, dim3(256), 0, st, sp);
      else if (WN == 4) hipLaunchKernelGGL((k_conv3_slab<4, true, 2>), grid, dim3(256), 0, st, sp);
      else hipLaunchKernelGGL((k_conv3_slab<2, true, 2>), grid, dim3(256), 0, st, sp);
    } else if (WN == 5) hipLaunchKernelGGL((k_conv3_slab<5, false, 2>), grid, dim3(256), 0, st, sp);
    else if (WN == 4) hipLaunchKernelGGL((k_conv3_slab<4, false, 2>), grid, dim3(256), 0, st, sp);
    else hipLaunchKernelGGL((k_conv3_slab<2, false, 2>), grid, dim3(256), 0, st, sp);
  } else if (f16) {
    if (pp_env && WN == 5) hipLaunchKernelGGL((k_conv3_slab<5, true, 4, 1>), grid, dim3(512), 0, st, sp);
    else if (pp_env && WN == 4) hipLaunchKernelGGL((k_conv3_slab<4, true, 4, 1>), grid, dim3(512), 0, st, sp);
    else if (WN == 5) hipLaunchKernelGGL((k_conv3_slab<5, true>), grid, dim3(512), 0, st, sp);
    else if (WN == 4) hipLaunchKernelGGL((k_conv3_slab<4, true>), grid, dim3(512), 0, st, sp);
    else hipLaunchKernelGGL((k_conv3_slab<2, true>), grid, dim3(512), 0, st, sp);
  } else if (pp_env == 2 && WN == 5) hipLaunchKernelGGL((k_conv3_slab<5, false, 4, 2>), grid, dim3(512), 0, st, sp);
  else if (pp_env == 2 && WN == 4) hipLaunchKernelGGL((k_conv3_slab<4, false, 4, 2>), grid, dim3(512), 0, st, sp);
  else if (pp_env && WN == 5) hipLaunchKernelGGL((k_conv3_slab<5, false, 4, 1>), grid, dim3(512), 0, st, sp);
  else if (pp_env && WN == 4) hipLaunchKernelGGL((k_conv3_slab<4, false, 4, 1>), grid, dim3(512), 0, st, sp);
  else if (WN == 5) hipLaunchKernelGGL((k_conv3_slab<5>), grid, dim3(512), 0, st, sp);
  else if (WN == 4) hipLaunchKernelGGL((k_conv3_slab<4>), grid, dim3(512), 0, st, sp);
  else hipLaunchKernelGGL((k_conv3_slab<2>), grid, dim3(512), 0, st, sp);
#ifdef TFMQ_PHASE_TIMERS
  if (sp.p.dbg && getenv("TFMQ_PHASE_PRINT")) {
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> hb(static_cast<size_t>(grid.x) * 4);
    (void)hipMemcpy(hb.data(), dbuf, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    double kl = 0, ep = 0;
    for (unsigned i = 0; i < grid.x; ++i) {
      t0 = hb[i * 4] < t0 ? hb[i * 4] : t0;
      t1 = hb[i * 4 + 2] > t1 ? hb[i * 4 + 2] : t1;
      kl += double(hb[i * 4 + 1] - hb[i * 4]);
      ep += double(hb[i * 4 + 2] - hb[i * 4 + 1]);
    }
    {
      std::vector<unsigned long long> kb(static_cast<size_t>(grid.x) * 16);
      (void)hipMemcpy(kb.data(), dbuf3, kb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double ks[7] = {0, 0, 0, 0, 0, 0, 0}, es[3] = {0, 0, 0};
      for (unsigned i = 0; i < grid.x; ++i) {
        for (int q = 0; q < 7; ++q) ks[q] += double(kb[i * 16 + q]);
        for (int q = 0; q < 3; ++q) es[q] += double(kb[i * 16 + 8 + q]);
      }
      fprintf(stderr, "[slab %dx%dx%d Cin%d Cout%d] epilogue of wave 0, shader cycles per block: table + barriers %.0f, (i, j) loop %.0f, store drain %.0f\n",
              d.B, d.H, d.W, d.Cin, d.Cout, es[0] / grid.x, es[1] / grid.x, es[2] / grid.x);
      const double stn = ks[6] > 0 ? ks[6] : 1;
      if (pp_env && WN >= 4 && !half_m && !f16) {
        std::vector<unsigned long long> kb2(static_cast<size_t>(grid.x) * 16);
        (void)hipMemcpy(kb2.data(), dbuf3 + static_cast<size_t>(65536) * 16, kb2.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double k2[7] = {0, 0, 0, 0, 0, 0, 0};
        for (unsigned i = 0; i < grid.x; ++i)
          for (int q = 0; q < 7; ++q) k2[q] += double(kb2[i * 16 + q]);
        const double s2 = k2[6] > 0 ? k2[6] : 1;
        fprintf(stderr, "[slab %dx%dx%d Cin%d Cout%d] PING-PONG step, shader cycles, wave 0 | wave 4: reads + DMA issue %.0f | %.0f, vmcnt wait %.0f | %.0f, barrier (load phase) %.0f | %.0f, MFMA issue %.0f | %.0f, barrier (compute phase) %.0f | %.0f = %.0f | %.0f per step\n",
                d.B, d.H, d.W, d.Cin, d.Cout, ks[0] / stn, k2[0] / s2, ks[1] / stn, k2[1] / s2, ks[2] / stn, k2[2] / s2, ks[3] / stn, k2[3] / s2, ks[4] / stn, k2[4] / s2,
                (ks[0] + ks[1] + ks[2] + ks[3] + ks[4]) / stn, (k2[0] + k2[1] + k2[2] + k2[3] + k2[4]) / s2);
      } else
      fprintf(stderr, "[slab %dx%dx%d Cin%d Cout%d] K-step of wave 0, shader cycles: vmcnt wait %.0f, barrier %.0f, fragment reads %.0f, MFMA issue (first half) %.0f, DMA issue %.0f, MFMA issue (second half) %.0f = %.0f per step\n",
              d.B, d.H, d.W, d.Cin, d.Cout, ks[0] / stn, ks[1] / stn, ks[2] / stn, ks[3] / stn, ks[4] / stn, ks[5] / stn, (ks[0] + ks[1] + ks[2] + ks[3] + ks[4] + ks[5]) / stn);
    }
    // how synchronised are the blocks: histogram of epilogue-start times over the launch span, 20 bins
    int hist[20] = {0};
    for (unsigned i = 0; i < grid.x; ++i) hist[int(double(hb[i * 4 + 1] - t0) / double(t1 - t0 + 1) * 20)]++;
    fprintf(stderr, "[slab %dx%dx%d Cin%d Cout%d res%d up%d] blocks %u: K loop %.2f us, epilogue %.2f us per block; span %.1f us; epilogue starts per 5%% of the span:",
            d.B, d.H, d.W, d.Cin, d.Cout, d.residual ? 1 : 0, d.up2x, grid.x, kl / grid.x / 100, ep / grid.x / 100, double(t1 - t0) / 100);
    for (int k = 0; k < 20; ++k) fprintf(stderr, " %d", hist[k]);
    fprintf(stderr, "\n");
  }
#endif
  return true;
}
