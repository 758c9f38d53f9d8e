// mg_inflate.hip — gzip / BGZF inflated ON THE DEVICE: compressed bytes cross PCIe, the text is born in HBM.
//
// The reference hands `.fq.gz` straight to kmc (scripts/select_db.py:50-52,146-148) and `zcat`s the selected genomes
// (:101-105).  Round 4 inflated on host threads (mg_pgzip.hip: 2.5e7 reads/s, 3.16 GB of text over the link for 0.52 GB of
// file).  Here the decoder of mg_inflate_core.h runs as one WAVEFRONT per job:
//
//   BGZF      every block says how long it is and how long its text is (the host walks the 18-byte headers): one job per block,
//             bytes written straight to their place, CRC-32 per block on the device.
//   gzip      the stream is cut into chunks of compressed bytes; k_find_block_starts finds in every chunk the first bit at which
//             a dynamic-Huffman block plausibly starts (64 bit positions per step and wavefront: header fields + the code-length
//             code's Kraft sum per lane, then the whole header by the wavefront); k_inflate<uint16_t> decodes from there to the
//             next chunk's start with an UNKNOWN window (16-bit symbols: a byte, or "byte i of the 32 KB in front of me"); the
//             host checks that every job ended exactly where the next one started (a job that started at a false positive is
//             dropped and the hole decoded again); k_window_chain hands the 32 KB window from job to job (one workgroup, the
//             window in LDS); k_resolve_text turns symbols into bytes at their final place; k_crc_segments + the host check
//             every member's CRC-32 and ISIZE.
//
// A file goes through in STAGES of compressed bytes (as many as make one ROUND of jobs: 24 per CU — the decoder is bound by its own
// latency, so the jobs in flight are what counts; 6144 jobs = ~197 MB on 256 CUs), the next stage starting at the bit where the
// previous one ended, with its window.  The compressed bytes go up through page-locked slots filled by reader
// threads while earlier stages decode.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

#include "mg_inflate.h"
#include "mg_inflate_core.h"

namespace mg {
using namespace mgi;

// ---------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------
// chunk c (blockIdx.x / parts + 1) of the stage: absolute bits [grid0 + c * chunk_bits, + chunk_bits) below limit_bit, part
// blockIdx.x % parts of it.  grid0 and chunk_bits / parts are multiples of 64.  starts[blockIdx.x] = first plausible block start in
// the part behind min_bit, or ~0.
__global__ __launch_bounds__(64) void k_find_block_starts(const uint32_t* __restrict__ in, uint64_t nbytes, uint64_t grid0, uint64_t chunk_bits,
                                                          uint64_t min_bit, uint64_t limit_bit, uint64_t* __restrict__ starts, uint32_t* __restrict__ info,
                                                          int strict_, int parts) {
  const bool strict = strict_ != 0;
  __shared__ uint32_t cand[128];  // candidate positions (bits behind lo), ascending
  const int lane = (int)threadIdx.x;
  const uint64_t nwords = (nbytes + 3) / 4, nbits = nbytes * 8;
  // `parts` workgroups per chunk, a part of it each (the host takes the first part that found something)
  const uint64_t part_bits = chunk_bits / (uint64_t)parts;
  const uint64_t lo = grid0 + (uint64_t)(blockIdx.x / (uint32_t)parts + 1) * chunk_bits + (uint64_t)(blockIdx.x % (uint32_t)parts) * part_bits;
  uint64_t hi = lo + part_bits;
  if (hi > limit_bit) hi = limit_bit;
  uint64_t found = ~0ull;
  uint32_t ncand = 0, tried = 0, nsteps = 0;
  __shared__ uint32_t words[72];  // 2048 bit positions and what the last of them look at
  __shared__ uint16_t pre[2048];  // positions (behind q0) whose header FIELDS are possible, ascending: their code lengths 64 at a time
  // the candidates' whole headers, up to 64 at once, a lane each (light_validate: the decision of validate_block_start — the host
  // check holds the two against each other at every position — in registers, so that this kernel needs none of the decoder's LDS
  // and a stage's chunks are all resident at once)
  auto validate_queued = [&]() {
    __syncthreads();
    for (uint32_t base = 0; base < ncand && found == ~0ull; base += 64) {
      const uint32_t ci = base + (uint32_t)lane;
      const bool good = ci < ncand && light_validate(in, nbytes, lo + cand[ci < ncand ? ci : 0], strict);
      const uint64_t g = __ballot(good);
      tried += ncand - base < 64 ? ncand - base : 64;
      if (g) found = lo + cand[base + (uint32_t)__builtin_ctzll(g)];
    }
    __syncthreads();
    ncand = 0;
  };
  // `take` queued positions from pre[base]: the code-length code of each (a lane each) -> the candidates
  auto code_lengths_of_queued = [&](uint64_t q0, uint32_t base, uint32_t take) {
    const uint32_t off = pre[base + ((uint32_t)lane < take ? (uint32_t)lane : 0u)];
    const uint32_t i = off >> 5, sft = off & 31u;
    const uint64_t x0 = (uint64_t)words[i] | (uint64_t)words[i + 1] << 32, x1 = (uint64_t)words[i + 2] | (uint64_t)words[i + 3] << 32;
    const uint64_t blo = sft ? (x0 >> sft) | (x1 << (64 - sft)) : x0, bhi = x1 >> sft;
    const bool ok = (uint32_t)lane < take && probe_code_lengths(blo, bhi, strict);
    const uint64_t m = __ballot(ok);
    if (ok) cand[ncand + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint32_t)(q0 + off - lo);
    ncand += (uint32_t)__builtin_popcountll(m);
    if (ncand >= 64) validate_queued();
  };
  // positions a candidate may stand at: behind min_bit, below hi, its 80 bits of header inside the input
  const uint64_t p_lower = min_bit + 1, p_upper = nbits >= 80 ? (hi < nbits - 79 ? hi : nbits - 79) : 0;
  for (uint64_t q0 = lo; q0 < hi && found == ~0ull; q0 += 2048) {
    const uint64_t w0 = q0 >> 5;
    __syncthreads();
    words[lane] = w0 + (uint64_t)lane < nwords ? in[w0 + lane] : 0u;
    if (lane < 8) words[64 + lane] = w0 + 64 + (uint64_t)lane < nwords ? in[w0 + 64 + lane] : 0u;
    __syncthreads();
    nsteps += (uint32_t)(((hi - q0 < 2048 ? hi - q0 : 2048) + 63) / 64);
    // the header's fields at the lane's 32 positions, all at once (probe_fields_mask) -> the queue, in ascending order
    const uint64_t pb = q0 + 32ull * (uint64_t)lane;
    const uint32_t from = p_lower > pb ? (p_lower - pb < 32 ? (uint32_t)(p_lower - pb) : 32u) : 0u;
    const uint32_t to = p_upper > pb ? (p_upper - pb < 32 ? (uint32_t)(p_upper - pb) : 32u) : 0u;
    const uint32_t in_range = (to >= 32u ? ~0u : (1u << to) - 1u) & ~(from >= 32u ? ~0u : (1u << from) - 1u);
    uint32_t mask = probe_fields_mask((uint64_t)words[lane] | (uint64_t)words[lane + 1] << 32) & in_range;
    const uint32_t mine = (uint32_t)__builtin_popcount(mask);
    const uint32_t incl = DevExec::dpp_inclusive(mine);
    const uint32_t npre = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t w = incl - mine;
    while (mask) {
      pre[w++] = (uint16_t)(32u * (uint32_t)lane + (uint32_t)__builtin_ctz(mask));
      mask &= mask - 1u;
    }
    __syncthreads();
    for (uint32_t base = 0; base < npre && found == ~0ull; base += 64) code_lengths_of_queued(q0, base, npre - base < 64 ? npre - base : 64u);
  }
  if (ncand && found == ~0ull) validate_queued();
  if (lane == 0) {
    starts[blockIdx.x] = found;
    info[2 * blockIdx.x] = tried;
    info[2 * blockIdx.x + 1] = nsteps;
  }
}

template <class OutT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_inflate(const uint32_t* __restrict__ in, uint64_t nbytes, int input_final, const Job* __restrict__ jobs,
                                                OutT* out, Result* __restrict__ results, Event* events, uint32_t* nevents, uint32_t max_events,
                                                uint16_t* tails) {
  __shared__ Shared sh;
  DevExec ex{&sh, (int)threadIdx.x};
  const uint32_t j = blockIdx.x;
  Job job;
  job.start_bit = jobs[j].start_bit;
  job.stop_bit = jobs[j].stop_bit;
  job.out_off = jobs[j].out_off;
  job.out_cap = jobs[j].out_cap;
  job.flags = jobs[j].flags;
  job.pad = 0;
  run_job<DevExec, OutT>(ex, sh, in, nbytes, input_final != 0, job, j, out + job.out_off, &results[j], events, nevents, max_events,
                         tails ? tails + (uint64_t)j * kWindow : nullptr);
}

// One job per LANE (mg_inflate_core.h, LaneDec): 64 serial decoders side by side, their tables in LDS (80 KB per wavefront), rounds
// until every lane is done.  What a file offers in deflate blocks is all in flight at once.
template <class OutT>
__global__ __launch_bounds__(64) void k_inflate_lanes(const uint32_t* __restrict__ in, uint64_t nbytes, int input_final, const Job* __restrict__ jobs,
                                                      uint32_t njobs, OutT* out, Result* __restrict__ results, Event* events, uint32_t* nevents,
                                                      uint32_t max_events, LaneScratch* scratch) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  LaneMem* mem = reinterpret_cast<LaneMem*>(lds_raw) + threadIdx.x;
  const uint32_t j = blockIdx.x * 64u + threadIdx.x;
  LaneDec<OutT> d;
  d.state = LS_DONE;
  if (j < njobs) {
    Job job;
    job.start_bit = jobs[j].start_bit;
    job.stop_bit = jobs[j].stop_bit;
    job.out_off = jobs[j].out_off;
    job.out_cap = jobs[j].out_cap;
    job.flags = jobs[j].flags;
    job.pad = 0;
    d.init(in, nbytes, input_final != 0, job, j, out + job.out_off, mem, scratch + j, events, nevents, max_events);
  }
  while (__any(d.state != LS_DONE)) {
    if (d.state != LS_DONE) d.round();
  }
  if (j < njobs) d.result(&results[j]);
}

// the tail rows of jobs decoded by k_inflate_lanes (the wavefront-per-job kernel writes its own): the window behind a job as far as the
// job knows it, one workgroup per job
__global__ __launch_bounds__(256) void k_make_tails(const Job* __restrict__ jobs, const Result* __restrict__ results, const uint16_t* __restrict__ sym,
                                                    uint16_t* __restrict__ tails) {
  const uint32_t j = blockIdx.x;
  if (results[j].overflow || results[j].status >= ST_ERR) return;
  const uint64_t n = results[j].out_count;
  const uint16_t* s = sym + jobs[j].out_off;
  uint16_t* row = tails + (uint64_t)j * kWindow;
  for (uint32_t w = threadIdx.x; w < kWindow; w += 256)
    row[w] = n >= kWindow ? s[n - kWindow + w] : (w < kWindow - n ? (uint16_t)(0x8000u | (w + (uint32_t)n)) : s[w - (kWindow - (uint32_t)n)]);
}

// The window from job to job.  Every job has left the window behind it AS FAR AS IT KNOWS IT (tail: 32768 symbols, window
// symbols where it depends on what was in front of the job).  Resolving that is a chain through all jobs; it is cut into
// groups of K jobs: k_chain_groups (a workgroup per group, the previous row in LDS) composes the rows of a group relative to the
// window in front of the GROUP; k_chain_tops (one workgroup) resolves the last row of every group from group to group;
// k_chain_rows resolves every row against its group's incoming window, all at once.  wins: (njobs + 1) rows of 32 KB bytes, row j =
// the window in front of job j (row 0 given).
struct ChainJob { const uint16_t* sym; const uint16_t* tail; uint64_t n; uint64_t t_off; uint64_t tile0; };
struct alignas(16) U16x8 { uint16_t v[8]; };
struct alignas(16) U8x16 { uint8_t v[16]; };

__global__ __launch_bounds__(1024) void k_chain_groups(const ChainJob* __restrict__ jobs, uint32_t njobs, uint32_t K, uint16_t* __restrict__ comp) {
  extern __shared__ uint16_t L16[];  // 2 x 32768 symbols
  const uint32_t t = threadIdx.x;
  const uint32_t j0 = blockIdx.x * K;
  uint32_t j1 = j0 + K;
  if (j1 > njobs) j1 = njobs;
  uint16_t* prev = L16;
  uint16_t* now = L16 + kWindow;
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) {
    U16x8 x;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) x.v[k] = (uint16_t)(0x8000u | (32u * t + 8u * i + k));
    *reinterpret_cast<U16x8*>(prev + 32u * t + 8u * i) = x;
  }
  U16x8 cur[4], nxt[4];
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) cur[i] = *reinterpret_cast<const U16x8*>(jobs[j0].tail + 32u * t + 8u * i);
  __syncthreads();
  for (uint32_t j = j0; j < j1; ++j) {
    if (j + 1 < j1) {
#pragma unroll
      for (uint32_t i = 0; i < 4; ++i) nxt[i] = *reinterpret_cast<const U16x8*>(jobs[j + 1].tail + 32u * t + 8u * i);
    }
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
#pragma unroll
      for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t v = cur[i].v[k];
        if (v & 0x8000u) cur[i].v[k] = prev[v & 0x7fffu];
      }
      *reinterpret_cast<U16x8*>(now + 32u * t + 8u * i) = cur[i];
      *reinterpret_cast<U16x8*>(comp + (uint64_t)j * kWindow + 32u * t + 8u * i) = cur[i];
    }
    __syncthreads();
    uint16_t* sw = prev; prev = now; now = sw;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) cur[i] = nxt[i];
  }
}

// tops: (G + 1) rows of bytes; row 0 = the window in front of the stage (given), row g + 1 = behind group g
__global__ __launch_bounds__(1024) void k_chain_tops(const uint16_t* __restrict__ comp, uint32_t njobs, uint32_t K, uint32_t G, uint8_t* tops) {
  extern __shared__ uint8_t L8[];  // 2 x 32 KB
  const uint32_t t = threadIdx.x;
  *reinterpret_cast<U8x16*>(L8 + 32u * t) = *reinterpret_cast<const U8x16*>(tops + 32u * t);
  *reinterpret_cast<U8x16*>(L8 + 32u * t + 16u) = *reinterpret_cast<const U8x16*>(tops + 32u * t + 16u);
  auto last = [&](uint32_t g) { const uint32_t e = (g + 1) * K; return (e < njobs ? e : njobs) - 1u; };
  U16x8 cur[4], nxt[4];
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) cur[i] = *reinterpret_cast<const U16x8*>(comp + (uint64_t)last(0) * kWindow + 32u * t + 8u * i);
  __syncthreads();
  for (uint32_t g = 0; g < G; ++g) {
    if (g + 1 < G) {
#pragma unroll
      for (uint32_t i = 0; i < 4; ++i) nxt[i] = *reinterpret_cast<const U16x8*>(comp + (uint64_t)last(g + 1) * kWindow + 32u * t + 8u * i);
    }
    const uint8_t* prev = L8 + (g & 1u) * kWindow;
    uint8_t* now = L8 + ((g + 1) & 1u) * kWindow;
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
      U8x16 r;
#pragma unroll
      for (uint32_t k = 0; k < 16; ++k) {
        const uint32_t v = cur[2 * h + (k >> 3)].v[k & 7u];
        r.v[k] = v & 0x8000u ? prev[v & 0x7fffu] : (uint8_t)v;
      }
      *reinterpret_cast<U8x16*>(now + 32u * t + 16u * h) = r;
      *reinterpret_cast<U8x16*>(tops + (uint64_t)(g + 1) * kWindow + 32u * t + 16u * h) = r;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) cur[i] = nxt[i];
  }
}

// wins row j + 1 = comp row j resolved against the window in front of j's group; one thread makes 16 bytes
__global__ __launch_bounds__(256) void k_chain_rows(const uint16_t* __restrict__ comp, uint32_t njobs, uint32_t K, const uint8_t* __restrict__ tops,
                                                    uint8_t* __restrict__ wins) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;  // 16-byte piece
  const uint64_t j = i / (kWindow / 16);
  if (j >= njobs) return;
  const uint32_t w0 = (uint32_t)(i % (kWindow / 16)) * 16u;
  const uint8_t* top = tops + (uint64_t)(j / K) * kWindow;
  const U16x8 a = *reinterpret_cast<const U16x8*>(comp + j * kWindow + w0), b = *reinterpret_cast<const U16x8*>(comp + j * kWindow + w0 + 8);
  U8x16 r;
#pragma unroll
  for (uint32_t k = 0; k < 16; ++k) {
    const uint32_t v = k < 8 ? a.v[k] : b.v[k - 8];
    r.v[k] = v & 0x8000u ? top[v & 0x7fffu] : (uint8_t)v;
  }
  *reinterpret_cast<U8x16*>(wins + (j + 1) * kWindow + w0) = r;
}

// symbols -> bytes at their final place.  A block makes one tile of 4096 text bytes of one job (tiles on a 16-byte grid of the
// text, so that full tiles leave as 16-byte stores); jobs[j].tile0 = the first tile of job j.
__global__ __launch_bounds__(256) void k_resolve_text(const ChainJob* __restrict__ jobs, uint32_t njobs, const uint8_t* __restrict__ wins,
                                                      uint8_t* __restrict__ text) {
  const uint32_t t = threadIdx.x;
  const uint64_t b = blockIdx.x;
  uint32_t lo = 0, hi = njobs;  // last job with tile0 <= b
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) / 2;
    if (jobs[mid].tile0 <= b) lo = mid; else hi = mid;
  }
  const uint32_t j = lo;
  const uint64_t t_off = jobs[j].t_off, n = jobs[j].n;
  const uint16_t* sym = jobs[j].sym;
  const uint8_t* win = wins + (uint64_t)j * kWindow;
  const uint64_t T0 = (t_off & ~15ull) + (b - jobs[j].tile0) * 4096ull;
  // a thread makes 16 consecutive bytes of the text (one aligned 16-byte store) from 16 consecutive symbols (two 16-byte loads,
  // wherever the job's symbols happen to start), window symbols looked up in the job's row of windows
  const uint64_t e0 = T0 + 16ull * t;
  if (e0 >= t_off && e0 + 16 <= t_off + n) {
    U16x8 a, c;
    __builtin_memcpy(&a, sym + (e0 - t_off), 16);
    __builtin_memcpy(&c, sym + (e0 - t_off) + 8, 16);
    U8x16 r;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
      const uint32_t v = a.v[i], w = c.v[i];
      r.v[i] = v < 256u ? (uint8_t)v : win[v & 0x7fffu];
      r.v[8 + i] = w < 256u ? (uint8_t)w : win[w & 0x7fffu];
    }
    *reinterpret_cast<U8x16*>(text + e0) = r;
  } else {
    for (uint32_t i = 0; i < 16; ++i) {
      const uint64_t e = e0 + i;
      if (e < t_off || e >= t_off + n) continue;
      const uint32_t v = sym[e - t_off];
      text[e] = v < 256u ? (uint8_t)v : win[v & 0x7fffu];
    }
  }
}

// CRC-32 of segments of the text, one wavefront per segment, COALESCED: a CRC register that starts at 0 is linear in the message, so
// lane l takes every 64th word (the words 4 l bytes into every 256-byte row) — per row one 256-byte load of the wavefront and, per
// lane, b = shift256(b) ^ word, the shift by 256 zero bytes being four table look-ups (tables made by the block) — and at the end
// a lane's register is shifted by what lies behind its last word and the 64 are XORed.  Rows are aligned to the END of the segment's
// whole words (leading zeros do not change a register that starts at 0); the at most three bytes in front of the first aligned word
// and behind the last one are lane 0's.  The initial and final inversions are put in at the end.
// (Rounds 1-2 of this kernel gave a lane a contiguous 4 KB: 64 cache lines per load instruction, 216 GB/s.)
struct CrcSeg { uint64_t start, len; };
struct X2N { uint32_t v[32]; };
__global__ __launch_bounds__(256) void k_crc_segments(const uint8_t* __restrict__ text, const CrcSeg* __restrict__ segs, uint32_t nsegs,
                                                      uint32_t* __restrict__ crcs, X2N x2n) {
  __shared__ uint32_t T0[256];
  __shared__ uint32_t TX[4][256];
  const uint32_t t = threadIdx.x;
  {
    uint32_t c = t;
    for (int k = 0; k < 8; ++k) c = c & 1u ? (c >> 1) ^ kCrcPoly : c >> 1;
    T0[t] = c;
    const uint32_t x256 = crc_x8n(256, x2n.v);
    for (int k = 0; k < 4; ++k) TX[k][t] = crc_multmodp(x256, t << (8 * k));
    __syncthreads();
  }
  const uint32_t seg = blockIdx.x * 4u + (t >> 6), lane = t & 63u;
  if (seg >= nsegs) return;
  const uint64_t start = segs[seg].start, len = segs[seg].len, end = start + len;
  const uint64_t base = reinterpret_cast<uintptr_t>(text);
  uint64_t A = start + ((4 - ((base + start) & 3u)) & 3u);  // the first aligned word
  if (A > end) A = end;
  const uint64_t M4 = (end - A) & ~3ull, K = (M4 + 255) / 256;
  const uint64_t mid_end = A + M4;
  uint32_t b = 0;
  for (uint64_t k = 0; k < K; ++k) {
    const int64_t at = (int64_t)mid_end - (int64_t)(256 * (K - k)) + 4 * (int64_t)lane;
    const uint32_t w = at >= (int64_t)A ? *reinterpret_cast<const uint32_t*>(text + at) : 0u;
    b = TX[3][b >> 24] ^ TX[2][(b >> 16) & 0xffu] ^ TX[1][(b >> 8) & 0xffu] ^ TX[0][b & 0xffu] ^ w;
  }
  // the register behind a lane's last word: four more bytes through the register, then what lies behind that word
  uint32_t c = b;
  for (int k = 0; k < 4; ++k) c = T0[c & 0xffu] ^ (c >> 8);
  const uint64_t tail = end - mid_end;
  c = K ? crc_multmodp(crc_x8n(4 * (63 - lane) + tail, x2n.v), c) : 0u;
  for (uint32_t s2 = 1; s2 < 64; s2 <<= 1) c ^= (uint32_t)__shfl_xor((int)c, (int)s2, 64);
  if (lane == 0) {
    uint32_t h = 0;  // the bytes in front of the first aligned word ...
    for (uint64_t i = start; i < A; ++i) h = T0[(h ^ text[i]) & 0xffu] ^ (h >> 8);
    uint32_t tl = 0;  // ... and behind the last one
    for (uint64_t i = mid_end; i < end; ++i) tl = T0[(tl ^ text[i]) & 0xffu] ^ (tl >> 8);
    const uint32_t raw = crc_multmodp(crc_x8n(end - A, x2n.v), h) ^ c ^ tl;
    crcs[seg] = raw ^ crc_multmodp(crc_x8n(len, x2n.v), 0xffffffffu) ^ 0xffffffffu;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// configuration (mg_inflate_config) and counters (mg_inflate_stats)
// ---------------------------------------------------------------------------------------------------------------------
struct InflateConfig {
  uint64_t chunk_bytes = 32u << 10;   // compressed bytes per job of a gzip stream
  // compressed bytes per stage.  0 = sized by the device: jobs take about the same time each, so a launch runs in ROUNDS of as many
  // jobs as the chip holds at once (24 per CU, bounded by LDS and registers: 6144 on 256 CUs) and a stage of 1.4 rounds costs two — a stage is
  // what is left of the file cut into equal parts of at most one round of jobs, at the jobs per byte the stages before it had
  uint64_t stage_bytes = 0;
  uint32_t ratio = 10;                // symbols reserved per compressed byte of a job (a job that needs more is decoded again)
  // launches of at least this many jobs decode a job per LANE (k_inflate_lanes) instead of a job per wavefront.  OFF by default: measured
  // (profiles/r05/inflate_lanes.txt) a round of the 64 side-by-side decoders takes ~3.5 us whatever the number of jobs — every lane's
  // loads and stores are its own cache lines — so 16 000 jobs (a 10M-read FASTQ) take 89 ms against 67 ms a job per wavefront;
  // it wins only from ~25 000 jobs in one launch
  uint32_t lane_jobs = 0xffffffffu;
  int on = 1;                         // .gz files of the streaming entry points take the device inflater
};
static InflateConfig g_cfg;
constexpr uint64_t kFindParts = 4;  // workgroups of k_find_block_starts per chunk
static mg_inflate_counters g_cnt;
static uint64_t round_of_jobs() {  // jobs the device decodes at once (a job per wavefront)
  static uint64_t v = 0;
  if (!v) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_inflate<uint16_t>), 64, 0) != hipSuccess || per_cu < 1) per_cu = 8;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 2048;
    v = (uint64_t)per_cu * (uint64_t)prop.multiProcessorCount;
  }
  return v;
}
InflateConfig& inflate_cfg() { return g_cfg; }
bool inflate_dev_enabled() { return g_cfg.on != 0; }

void inflate_release_all();
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---------------------------------------------------------------------------------------------------------------------
// the compressed bytes: file -> page-locked slots (reader threads) -> device, in order, while the stages run
// ---------------------------------------------------------------------------------------------------------------------
struct CompUploader {
  static constexpr uint64_t kPiece = 16ull << 20;
  static constexpr size_t kSlots = 6;
  struct Piece { const uint8_t* src; uint8_t* dst; uint64_t len, end; };  // end: bytes of the whole job up to and with this piece
  std::vector<Piece> pieces;
  uint64_t n = 0;
  uint64_t npieces = 0;
  hipStream_t copy = nullptr;
  std::vector<uint8_t*> slots;
  std::vector<hipEvent_t> ev_piece;  // piece i is on the device
  std::mutex m;
  std::condition_variable cv;
  std::vector<int> filled;           // piece i is in its slot
  uint64_t queued = 0;               // pieces whose DMA has been queued
  std::atomic<uint64_t> next{0};
  bool stop = false, failed = false;
  std::vector<std::thread> readers;
  std::thread dma;
  int device = 0;

  ~CompUploader() { finish(); }
  // page-locking a slot costs milliseconds: the slots and the copy stream stay with the library (inflate_release_all)
  // (two sets: 0 = the inflater's compressed bytes, 1 = table uploads that run BESIDE a stream — mg_refdb_upload_begin)
  // A set serves ONE job at a time (`busy`): a second job that starts while the first is still copying — two table uploads in
  // flight, a table upload beside a `.gz` stream on the same set — takes slots and a copy stream of its own and gives them back
  // when it ends (their reader threads would otherwise fill the same slots and mix the two uploads).
  struct Kept { std::vector<uint8_t*> slots; hipStream_t copy = nullptr; std::atomic<bool> busy{false}; };
  static Kept& kept(int set = 0) { static Kept k[2]; return k[set & 1]; }
  Kept* holds = nullptr;             // the set this job has taken
  std::vector<uint8_t*> own_slots;   // ... or its own slots and stream
  hipStream_t own_copy = nullptr;
  void finish() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : readers) if (t.joinable()) t.join();
    if (dma.joinable()) dma.join();
    readers.clear();
    if (copy) (void)hipStreamSynchronize(copy);
    copy = nullptr;
    slots.clear();
    if (holds) { holds->busy.store(false); holds = nullptr; }
    for (uint8_t* p : own_slots) (void)hipHostFree(p);
    own_slots.clear();
    if (own_copy) { (void)hipStreamDestroy(own_copy); own_copy = nullptr; }
    for (hipEvent_t e : ev_piece) (void)hipEventDestroy(e);
    ev_piece.clear();
  }
  int start(const uint8_t* s, uint64_t nbytes, uint8_t* d, int nthreads) {
    std::vector<std::pair<const void*, std::pair<void*, uint64_t>>> one{{s, {d, nbytes}}};
    return start_ranges(one, nthreads);
  }
  // several host arrays, each to its own place on the device, as one job
  int start_ranges(const std::vector<std::pair<const void*, std::pair<void*, uint64_t>>>& ranges, int nthreads, int set = 0) {
    device = ctx().device;
    pieces.clear();
    n = 0;
    for (const auto& r : ranges)
      for (uint64_t at = 0; at < r.second.second; at += kPiece) {
        const uint64_t len = std::min(kPiece, r.second.second - at);
        n += len;
        pieces.push_back(Piece{static_cast<const uint8_t*>(r.first) + at, static_cast<uint8_t*>(r.second.first) + at, len, n});
      }
    npieces = pieces.size();
    if (npieces == 0) return MG_OK;
    Kept& k = kept(set);
    const size_t ns = npieces < kSlots ? (size_t)npieces : kSlots;
    if (!k.busy.exchange(true)) {
      holds = &k;
      if (!k.copy) MG_HIP(hipStreamCreateWithFlags(&k.copy, hipStreamNonBlocking));
      copy = k.copy;
      while (k.slots.size() < ns) {
        uint8_t* p = nullptr;
        MG_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), kPiece, hipHostMallocDefault));
        k.slots.push_back(p);
      }
      slots.assign(k.slots.begin(), k.slots.begin() + (long)ns);
    } else {  // the set is another job's for now
      MG_HIP(hipStreamCreateWithFlags(&own_copy, hipStreamNonBlocking));
      copy = own_copy;
      while (own_slots.size() < ns) {
        uint8_t* p = nullptr;
        MG_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), kPiece, hipHostMallocDefault));
        own_slots.push_back(p);
      }
      slots = own_slots;
    }
    for (uint64_t i = 0; i < npieces; ++i) {
      hipEvent_t e = nullptr;
      MG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ev_piece.push_back(e);
    }
    filled.assign(npieces, 0);
    if (nthreads < 1) nthreads = 1;
    if ((uint64_t)nthreads > ns) nthreads = (int)ns;
    for (int t = 0; t < nthreads; ++t) readers.emplace_back([this] { read_loop(); });
    dma = std::thread([this] { dma_loop(); });
    return MG_OK;
  }
  void read_loop() {
    for (;;) {
      const uint64_t i = next.fetch_add(1);
      if (i >= npieces) return;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || i < queued + slots.size(); });  // the slot's previous piece has been queued ...
        if (stop) return;
      }
      if (i >= slots.size()) (void)hipEventSynchronize(ev_piece[i - slots.size()]);  // ... and has left it
      memcpy(slots[i % slots.size()], pieces[i].src, pieces[i].len);
      {
        std::lock_guard<std::mutex> lk(m);
        filled[i] = 1;
      }
      cv.notify_all();
    }
  }
  void dma_loop() {
    (void)hipSetDevice(device);
    for (uint64_t i = 0; i < npieces; ++i) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || filled[i]; });
        if (stop) return;
      }
      bool ok = hipMemcpyAsync(pieces[i].dst, slots[i % slots.size()], pieces[i].len, hipMemcpyHostToDevice, copy) == hipSuccess;
      ok = ok && hipEventRecord(ev_piece[i], copy) == hipSuccess;
      {
        std::lock_guard<std::mutex> lk(m);
        if (!ok) failed = true;
        queued = i + 1;
      }
      cv.notify_all();
      if (!ok) return;
    }
  }
  // makes `st` wait until the first `bytes` bytes of the job are on the device
  int need(uint64_t bytes, hipStream_t st) {
    if (bytes == 0 || npieces == 0) return MG_OK;
    if (bytes > n) bytes = n;
    uint64_t last = 0;
    while (pieces[last].end < bytes) ++last;
    {
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return failed || queued > last; });
      if (failed) return fail(MG_ERR_HIP, "uploading through the page-locked slots failed");
    }
    MG_HIP(hipStreamWaitEvent(st, ev_piece[last], 0));
    return MG_OK;
  }
};

// Host arrays (pageable memory: memory maps of a stored table) -> the device through the library's page-locked slots, reader threads
// copying into the slots while earlier pieces are on the wire; `st` waits for the last piece; returns when everything is there.
// (hipMemcpyAsync from pageable memory stages through the runtime's own small buffers: the reference-pipeline table's 440 MB took 25 ms.)
int upload_ranges(const std::vector<std::pair<const void*, std::pair<void*, uint64_t>>>& ranges, hipStream_t st) {
  // the destinations are blocks the caller has just taken from the pool — where a kernel of the main stream may still be reading
  // or writing them (a block freed there is handed out again at once): the copy stream writes only after the main stream's work
  MG_HIP(hipStreamSynchronize(ctx().stream));
  CompUploader up;
  MG_TRY(up.start_ranges(ranges, 4));
  MG_TRY(up.need(up.n, st));
  up.finish();
  return MG_OK;
}

// ... in two halves: begin returns with the reader threads and the DMA thread at work (the caller's arrays must stay where they are
// until end), end makes `st` wait for the last piece and joins them.  A set of slots of its own: a `.gz` reads file may be going up
// through the inflater's at the same time.
struct UploadJob { CompUploader up; };
// the jobs begun and not ended: mg_shutdown ends them before the library's slots go (inflate_release_all)
static std::mutex g_jobs_m;
static std::vector<UploadJob*> g_jobs;
static void job_forget(UploadJob* job) {
  std::lock_guard<std::mutex> lk(g_jobs_m);
  g_jobs.erase(std::remove(g_jobs.begin(), g_jobs.end(), job), g_jobs.end());
}
int upload_ranges_begin(const std::vector<std::pair<const void*, std::pair<void*, uint64_t>>>& ranges, UploadJob** out) {
  MG_HIP(hipStreamSynchronize(ctx().stream));  // (as upload_ranges: the destinations may be blocks the main stream has not let go of)
  std::unique_ptr<UploadJob> job(new UploadJob());
  MG_TRY(job->up.start_ranges(ranges, 4, 1));
  {
    std::lock_guard<std::mutex> lk(g_jobs_m);
    g_jobs.push_back(job.get());
  }
  *out = job.release();
  return MG_OK;
}
int upload_ranges_end(UploadJob* job, hipStream_t st) {
  if (!job) return MG_OK;
  std::unique_ptr<UploadJob> hold(job);
  job_forget(job);
  MG_TRY(job->up.need(job->up.n, st));
  job->up.finish();
  return MG_OK;
}
void upload_ranges_abort(UploadJob* job) {  // (~CompUploader joins the threads and waits for what is on the wire)
  if (job) job_forget(job);
  delete job;
}

void inflate_release_all() {
  {  // uploads still pending (a table handle that was never waited for): their threads stop here, before their slots are freed
    std::lock_guard<std::mutex> lk(g_jobs_m);
    for (UploadJob* j : g_jobs) j->up.finish();
    g_jobs.clear();
  }
  for (int set = 0; set < 2; ++set) {
    CompUploader::Kept& k = CompUploader::kept(set);
    if (k.copy) { (void)hipStreamSynchronize(k.copy); (void)hipStreamDestroy(k.copy); }
    for (uint8_t* p : k.slots) (void)hipHostFree(p);
    k.slots.clear();
    k.copy = nullptr;
    k.busy.store(false);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// the inflater
// ---------------------------------------------------------------------------------------------------------------------
struct BgzfBlock { uint64_t coff; uint32_t csize, isize, hdr; };

static bool bgzf_index(const uint8_t* h, uint64_t n, std::vector<BgzfBlock>* blocks) {
  uint64_t at = 0;
  while (at < n) {
    if (n - at < 28) return false;
    const uint8_t* p = h + at;
    if (!(p[0] == 0x1f && p[1] == 0x8b && p[2] == 8 && (p[3] & 4) && p[12] == 'B' && p[13] == 'C' && p[14] == 2 && p[15] == 0)) return false;
    if (p[3] != 4) return false;  // (only the extra field: what bgzip / htslib write)
    const uint32_t xlen = (uint32_t)(p[10] | (p[11] << 8));
    if (xlen != 6) return false;
    const uint32_t csize = (uint32_t)(p[16] | (p[17] << 8)) + 1u;
    if (csize < 26 || at + csize > n) return false;
    const uint8_t* tl = h + at + csize - 4;
    const uint32_t isize = (uint32_t)tl[0] | ((uint32_t)tl[1] << 8) | ((uint32_t)tl[2] << 16) | ((uint32_t)tl[3] << 24);
    if (isize > 65536u) return false;
    blocks->push_back(BgzfBlock{at, csize, isize, 12u + xlen});
    at += csize;
  }
  return true;
}

static const char* status_text(uint32_t st) {
  switch (st) {
    case ST_TRUNC: return "the gzip stream ends inside a member (truncated file)";
    case ST_BAD_BLOCK: return "corrupt deflate data: invalid block type";
    case ST_BAD_STORED: return "corrupt deflate data: invalid stored block lengths";
    case ST_BAD_CODES: return "corrupt deflate data: invalid code lengths set";
    case ST_BAD_SYMBOL: return "corrupt deflate data: invalid literal/length or distance code";
    case ST_BAD_DIST: return "corrupt deflate data: invalid distance too far back";
    case ST_BAD_HEADER: return "not in gzip format";
    case ST_EVENTS_FULL: return "more gzip members in one stage than the inflater keeps track of";
    default: return "unexpected decoder state";
  }
}

struct DevInflater {
  const uint8_t* h = nullptr;  // the compressed bytes on the host
  uint64_t n = 0;
  DevBuf comp;                 // ... and on the device (whole file)
  CompUploader up;
  hipStream_t st = nullptr;
  bool bgzf = false;
  std::vector<BgzfBlock> blocks;
  uint64_t next_block = 0;
  // gzip: where the next stage starts
  uint64_t next_bit = 0;
  bool first = true, done = false;
  DevBuf win;                  // the 32 KB in front of next_bit
  uint32_t mcrc = 0;           // the open member: CRC and length of what has been produced of it
  uint64_t mlen = 0;
  uint32_t x2n[32];
  std::vector<std::pair<uint64_t, uint32_t>> x8n_cache;  // (length, x^(8 length) mod p)

  uint64_t seen_jobs = 0, seen_bytes = 0;  // of the stages so far: the jobs a compressed byte makes
  uint64_t stage_size() const {
    if (g_cfg.stage_bytes) return g_cfg.stage_bytes;
    const uint64_t left = n - std::min<uint64_t>(n, next_bit / 8);
    const double per_byte = seen_bytes ? (double)seen_jobs / (double)seen_bytes : 1.0 / (double)g_cfg.chunk_bytes;
    const double room = 0.98 * (double)round_of_jobs();
    const uint64_t parts = std::max<uint64_t>(1, (uint64_t)std::ceil((double)left * per_byte / room));
    return std::max<uint64_t>((left + parts - 1) / parts, 1ull << 20);
  }
  uint64_t headroom = 0;  // bytes left free in front of every stage's text (the caller's carried record)
  int open(const uint8_t* host, uint64_t nbytes, int upload_threads, uint64_t headroom_) {
    h = host;
    n = nbytes;
    headroom = headroom_;
    // a stream of its own: a stage decodes while the caller parses and hashes the previous stage's text on the library's main stream
    Context& c = ctx();
    if (!c.stream_inf) MG_HIP(hipStreamCreateWithFlags(&c.stream_inf, hipStreamNonBlocking));
    st = c.stream_inf;
    c.inf_side = true;  // (blocks freed from now on are fenced across the streams: mg_core.hip, pool_free)
    crc_make_x2n(x2n);
    if (n < 2 || !(h[0] == 0x1f && h[1] == 0x8b)) return fail(MG_ERR_ARG, "%s", status_text(ST_BAD_HEADER));
    if (n < 18) return fail(MG_ERR_ARG, "%s", status_text(ST_TRUNC));
    bgzf = bgzf_index(h, n, &blocks);
    if (!bgzf) blocks.clear();
    MG_TRY(comp.alloc(((n + 3) & ~3ull) + 64));
    MG_TRY(win.alloc(kWindow));
    // Both blocks come out of the pool, where a block freed on the main stream is handed out again at once ("stream order is
    // enough" — for the main stream): what writes into them next are the copy stream and this one.  The main stream's work first.
    MG_HIP(hipStreamSynchronize(c.stream));
    // (the bytes behind the file's end inside its last word are read by nobody: BitReader::load is bounded by words, the decoder by bits)
    MG_HIP(hipMemsetAsync(comp.as<uint8_t>() + (n & ~3ull), 0, 64, st));
    MG_HIP(hipMemsetAsync(win.p, 0, kWindow, st));
    MG_HIP(hipStreamSynchronize(st));
    MG_TRY(up.start(h, n, comp.as<uint8_t>(), upload_threads));
    return MG_OK;
  }

  uint32_t x8n(uint64_t len) {
    for (auto& e : x8n_cache) if (e.first == len) return e.second;
    const uint32_t v = crc_x8n(len, x2n);
    if (x8n_cache.size() < 8) x8n_cache.emplace_back(len, v);
    return v;
  }

  static int lanes_attr() {  // (80 KB of dynamic LDS per wavefront: above the default limit)
    static bool done = false;
    if (!done) {
      MG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_inflate_lanes<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * sizeof(LaneMem)));
      MG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_inflate_lanes<uint8_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * sizeof(LaneMem)));
      done = true;
    }
    return MG_OK;
  }

  // CRC-32 of text[seg] for every segment -> host
  int crc_of(const uint8_t* d_text, const std::vector<CrcSeg>& segs, std::vector<uint32_t>* out) {
    out->assign(segs.size(), 0);
    if (segs.empty()) return MG_OK;
    DevBuf dsegs, dcrc;
    MG_TRY(dsegs.alloc(segs.size() * sizeof(CrcSeg)));
    MG_TRY(dcrc.alloc(segs.size() * 4));
    MG_HIP(hipMemcpyAsync(dsegs.p, segs.data(), segs.size() * sizeof(CrcSeg), hipMemcpyHostToDevice, st));
    X2N x;
    memcpy(x.v, x2n, sizeof(x2n));
    {
      ProfScope ps("k_crc_segments", st);
      k_crc_segments<<<(unsigned)((segs.size() + 3) / 4), 256, 0, st>>>(d_text, dsegs.as<CrcSeg>(), (uint32_t)segs.size(), dcrc.as<uint32_t>(), x);
    }
    MG_HIP(hipMemcpyAsync(out->data(), dcrc.p, segs.size() * 4, hipMemcpyDeviceToHost, st));
    MG_HIP(hipStreamSynchronize(st));
    return MG_OK;
  }

  // ---- BGZF: a stage = consecutive blocks of at most text_cap bytes of text ----
  struct PendingBz {
    uint64_t b0 = 0, b1 = 0, bytes = 0;
    DevBuf text, djobs, dres, dev, scr;
    std::vector<CrcSeg> segs;
  } pb;
  int bz_begin(uint64_t text_cap) {
    const uint64_t b0 = next_block;
    uint64_t b1 = b0, bytes = 0;
    while (b1 < blocks.size() && (b1 == b0 || bytes + blocks[b1].isize <= text_cap) && b1 - b0 < (1u << 20)) bytes += blocks[b1++].isize;
    const uint64_t nj = b1 - b0;
    pb.b0 = b0; pb.b1 = b1; pb.bytes = bytes;
    MG_TRY(pb.text.alloc(headroom + bytes + 64));
    next_block = b1;
    if (nj == 0) return MG_OK;
    std::vector<Job> jobs(nj);
    pb.segs.resize(nj);
    uint64_t at = 0;
    for (uint64_t i = 0; i < nj; ++i) {
      const BgzfBlock& b = blocks[b0 + i];
      jobs[i] = Job{(b.coff + b.hdr) * 8, ~0ull, at, b.isize, F_ONE_MEMBER | F_MEMBER_START, 0};
      pb.segs[i] = CrcSeg{headroom + at, b.isize};
      at += b.isize;
    }
    MG_TRY(up.need(blocks[b1 - 1].coff + blocks[b1 - 1].csize, st));
    DevBuf &djobs = pb.djobs, &dres = pb.dres, &dev = pb.dev;
    DevBuf* text = &pb.text;
    MG_TRY(djobs.alloc(nj * sizeof(Job)));
    MG_TRY(dres.alloc(nj * sizeof(Result)));
    MG_TRY(dev.alloc(64));
    MG_HIP(hipMemcpyAsync(djobs.p, jobs.data(), nj * sizeof(Job), hipMemcpyHostToDevice, st));
    MG_HIP(hipStreamSynchronize(st));  // (jobs is a local: the copy has to have read it)
    MG_HIP(hipMemsetAsync(dev.p, 0, 64, st));
    if (nj >= g_cfg.lane_jobs) {
      DevBuf& scr = pb.scr;
      MG_TRY(scr.alloc(nj * sizeof(LaneScratch)));
      MG_TRY(lanes_attr());
      ProfScope ps("k_inflate_bgzf", st);
      k_inflate_lanes<uint8_t><<<(unsigned)((nj + 63) / 64), 64, 64 * sizeof(LaneMem), st>>>(
          comp.as<uint32_t>(), n, 1, djobs.as<Job>(), (uint32_t)nj, text->as<uint8_t>() + headroom, dres.as<Result>(), nullptr, dev.as<uint32_t>(), 0,
          scr.as<LaneScratch>());
    } else {
      ProfScope ps("k_inflate_bgzf", st);
      k_inflate<uint8_t><<<(unsigned)nj, 64, 0, st>>>(comp.as<uint32_t>(), n, 1, djobs.as<Job>(), text->as<uint8_t>() + headroom, dres.as<Result>(),
                                                      nullptr, dev.as<uint32_t>(), 0, nullptr);
    }
    return MG_OK;
  }
  int bz_finish(DevBuf* text_out, uint64_t* ntext, bool* finished) {
    const uint64_t b0 = pb.b0, b1 = pb.b1, nj = b1 - b0;
    *ntext = pb.bytes;
    *finished = b1 >= blocks.size();
    if (nj) {
      std::vector<Result> res(nj);
      MG_HIP(hipMemcpyAsync(res.data(), pb.dres.p, nj * sizeof(Result), hipMemcpyDeviceToHost, st));
      std::vector<uint32_t> crcs;
      MG_TRY(crc_of(pb.text.as<uint8_t>(), pb.segs, &crcs));  // (synchronises: res is there as well)
      for (uint64_t i = 0; i < nj; ++i) {
        const BgzfBlock& b = blocks[b0 + i];
        const Result& r = res[i];
        if (r.status != ST_MEMBER) return fail(MG_ERR_ARG, "BGZF block at byte %llu: %s", (unsigned long long)b.coff, status_text(r.status));
        if (r.overflow || r.out_count != b.isize || r.isize != b.isize || r.end_bit != (b.coff + b.csize) * 8)
          return fail(MG_ERR_ARG, "BGZF block at byte %llu: corrupt (size mismatch)", (unsigned long long)b.coff);
        if (r.crc != crcs[i]) return fail(MG_ERR_ARG, "BGZF block at byte %llu: corrupt (CRC mismatch)", (unsigned long long)b.coff);
      }
      g_cnt.jobs += nj;
      for (const Result& r : res) {
        g_cnt.clk_tables += r.t_tab; g_cnt.clk_decode += r.t_dec; g_cnt.clk_emit += r.t_emit; g_cnt.clk_tail += r.t_tail;
        g_cnt.batches += r.nbatch; g_cnt.windows += r.nstep; g_cnt.blocks += r.nblocks; g_cnt.symbols_out += r.out_count;
      }
    } else {
      MG_HIP(hipStreamSynchronize(st));
    }
    g_cnt.stages += 1;
    *text_out = std::move(pb.text);
    pb = PendingBz();
    return MG_OK;
  }

  // ---- gzip ----
  struct Launch {  // jobs decoded together into one symbol buffer
    std::vector<Job> jobs;
    std::vector<Result> res;
    DevBuf sym, tails;  // tails: 32768 symbols per job, the window behind it as far as it knows it
    DevBuf djobs, dres, dev, dcnt, scr;
  };
  static constexpr uint32_t kMaxEvents = 1u << 16;
  // decode `jobs` (out_off / out_cap filled in here from `caps`) -> results and member ends
  int run_launch(Launch* L, const std::vector<uint64_t>& caps, bool input_final, uint64_t avail, std::vector<Event>* events, uint32_t launch_id) {
    MG_TRY(launch_jobs(L, caps, input_final, avail));
    return collect_jobs(L, events, launch_id);
  }
  // ... in two halves: everything up to the kernel (nothing waits) | the results
  int launch_jobs(Launch* L, const std::vector<uint64_t>& caps, bool input_final, uint64_t avail) {
    const size_t nj = L->jobs.size();
    uint64_t total = 0;
    for (size_t i = 0; i < nj; ++i) {
      L->jobs[i].out_off = total;
      L->jobs[i].out_cap = caps[i];
      total += (caps[i] + 7) & ~7ull;
    }
    MG_TRY(L->sym.alloc(total * 2 + 64));
    MG_TRY(L->tails.alloc((uint64_t)nj * kWindow * 2));
    DevBuf &djobs = L->djobs, &dres = L->dres, &dev = L->dev, &dcnt = L->dcnt;
    const uint32_t max_events = kMaxEvents;
    MG_TRY(djobs.alloc(nj * sizeof(Job)));
    MG_TRY(dres.alloc(nj * sizeof(Result)));
    MG_TRY(dev.alloc(max_events * sizeof(Event)));
    MG_TRY(dcnt.alloc(64));
    MG_HIP(hipMemcpyAsync(djobs.p, L->jobs.data(), nj * sizeof(Job), hipMemcpyHostToDevice, st));
    MG_HIP(hipMemsetAsync(dcnt.p, 0, 64, st));
    if (nj >= g_cfg.lane_jobs) {
      DevBuf& scr = L->scr;
      MG_TRY(scr.alloc(nj * sizeof(LaneScratch)));
      MG_TRY(lanes_attr());
      {
        ProfScope ps("k_inflate", st);
        k_inflate_lanes<uint16_t><<<(unsigned)((nj + 63) / 64), 64, 64 * sizeof(LaneMem), st>>>(
            comp.as<uint32_t>(), avail, input_final ? 1 : 0, djobs.as<Job>(), (uint32_t)nj, L->sym.as<uint16_t>(), dres.as<Result>(), dev.as<Event>(),
            dcnt.as<uint32_t>(), max_events, scr.as<LaneScratch>());
      }
      ProfScope ps("k_make_tails", st);
      k_make_tails<<<(unsigned)nj, 256, 0, st>>>(djobs.as<Job>(), dres.as<Result>(), L->sym.as<uint16_t>(), L->tails.as<uint16_t>());
    } else {
      ProfScope ps("k_inflate", st);
      k_inflate<uint16_t><<<(unsigned)nj, 64, 0, st>>>(comp.as<uint32_t>(), avail, input_final ? 1 : 0, djobs.as<Job>(), L->sym.as<uint16_t>(),
                                                       dres.as<Result>(), dev.as<Event>(), dcnt.as<uint32_t>(), max_events, L->tails.as<uint16_t>());
    }
    return MG_OK;
  }
  int collect_jobs(Launch* L, std::vector<Event>* events, uint32_t launch_id) {
    const size_t nj = L->jobs.size();
    DevBuf &dres = L->dres, &dev = L->dev, &dcnt = L->dcnt;
    const uint32_t max_events = kMaxEvents;
    L->res.resize(nj);
    uint32_t nev = 0;
    MG_HIP(hipMemcpyAsync(L->res.data(), dres.p, nj * sizeof(Result), hipMemcpyDeviceToHost, st));
    MG_HIP(hipMemcpyAsync(&nev, dcnt.p, 4, hipMemcpyDeviceToHost, st));
    MG_HIP(hipStreamSynchronize(st));
    if (nev > max_events) return fail(MG_ERR_CAPACITY, "%s", status_text(ST_EVENTS_FULL));
    if (nev) {
      const size_t at = events->size();
      events->resize(at + nev);
      MG_HIP(hipMemcpyAsync(events->data() + at, dev.p, nev * sizeof(Event), hipMemcpyDeviceToHost, st));
      MG_HIP(hipStreamSynchronize(st));
      for (size_t i = at; i < events->size(); ++i) (*events)[i].pad = launch_id;
    }
    g_cnt.jobs += nj;
    for (const Result& r : L->res) {
      g_cnt.clk_tables += r.t_tab; g_cnt.clk_decode += r.t_dec; g_cnt.clk_emit += r.t_emit; g_cnt.clk_tail += r.t_tail;
      g_cnt.batches += r.nbatch; g_cnt.windows += r.nstep; g_cnt.blocks += r.nblocks; g_cnt.symbols_out += r.out_count;
      for (int i = 0; i < 6; ++i) g_cnt.clk_sub[i] += r.t_sub[i];
    }
    return MG_OK;
  }

  // The text of the next stage: compressed bytes [next_bit / 8, ...) of which `avail` are on their way to the device (`final`:
  // that is the whole file).  *ntext = 0 with *finished = false: nothing could be finished with the bytes there are.
  // A stage in two halves.  gz_begin: compressed bytes [next_bit / 8, ...) of which `avail` are on their way to the device (`final`:
  // that is the whole file) -> block starts found (one wait), the jobs launched; returns with the decoder RUNNING on the inflater's
  // stream.  gz_finish: its results -> the stage's text; *ntext = 0 with *finished = false: nothing could be finished with the bytes
  // there were.
  struct PendingGz {
    bool active = false, nothing = false, final = false;
    uint64_t avail = 0, limit_bit = 0, scan_end = 0;
    double t_begin = 0, t_dec0 = 0;
    std::deque<Launch> launches;
  } pg;
  int gz_begin(uint64_t avail, bool final) {
    pg = PendingGz();
    pg.active = true;
    pg.avail = avail;
    pg.final = final;
    pg.t_begin = now_s();
    const uint64_t margin = 4ull << 20;
    const uint64_t chunk_bits = ((g_cfg.chunk_bytes + 8 * kFindParts - 1) / (8 * kFindParts)) * (64 * kFindParts);  // (every part on the 64-bit grid)
    uint64_t limit_bit = ~0ull;  // the last job stops at the first block boundary at or behind it
    {
      const uint64_t stage_end = (next_bit / 8 + stage_size()) * 8;
      const uint64_t safe_end = final ? avail * 8 : (avail > margin ? (avail - margin) * 8 : 0);
      if (!final || stage_end < safe_end) limit_bit = std::min(stage_end, safe_end);
      if (limit_bit != ~0ull && limit_bit <= next_bit + 64) {
        if (!final) { pg.nothing = true; return MG_OK; }  // more bytes first
        limit_bit = ~0ull;
      }
    }
    MG_TRY(up.need(avail, st));
    // 1. block starts
    const uint64_t grid0 = (next_bit / 64) * 64;
    const uint64_t scan_end = limit_bit == ~0ull ? avail * 8 : limit_bit;
    pg.limit_bit = limit_bit;
    pg.scan_end = scan_end;
    const uint64_t nchunks = scan_end > grid0 ? (scan_end - grid0 + chunk_bits - 1) / chunk_bits : 1;
    std::vector<uint64_t> starts;
    const double t_find0 = now_s();
    if (nchunks > 1) {
      DevBuf dstarts, dinfo;
      const uint64_t nparts = (nchunks - 1) * kFindParts;
      MG_TRY(dstarts.alloc(nparts * 8));
      MG_TRY(dinfo.alloc(nparts * 8));
      std::vector<uint64_t> found_at(nparts);
      std::vector<uint32_t> info(2 * nparts);
      starts.assign(nchunks - 1, ~0ull);
      // strict first (mg_inflate_core.h, probe_bits: only headers as encoders write them); a stage that finds next to nothing that
      // way — a writer that pads its length lists — is searched again by the format's rules alone
      for (int strict = dbg("inflate_loose_find") ? 0 : 1; strict >= 0; --strict) {
        {
          ProfScope ps("k_find_block_starts", st);
          k_find_block_starts<<<(unsigned)nparts, 64, 0, st>>>(comp.as<uint32_t>(), avail, grid0, chunk_bits, next_bit, scan_end, dstarts.as<uint64_t>(), dinfo.as<uint32_t>(), strict, (int)kFindParts);
        }
        MG_HIP(hipMemcpyAsync(found_at.data(), dstarts.p, nparts * 8, hipMemcpyDeviceToHost, st));
        MG_HIP(hipMemcpyAsync(info.data(), dinfo.p, nparts * 8, hipMemcpyDeviceToHost, st));
        MG_HIP(hipStreamSynchronize(st));
        uint64_t found = 0;
        for (uint64_t c = 0; c + 1 < nchunks; ++c) {  // a chunk's first start: of its first part that has one
          starts[c] = ~0ull;
          for (uint64_t q = 0; q < kFindParts && starts[c] == ~0ull; ++q) starts[c] = found_at[c * kFindParts + q];
          found += starts[c] != ~0ull;
        }
        if (!strict || nchunks - 1 < 8 || found * 4 >= nchunks - 1) break;
      }
      for (uint64_t c = 0; c < nparts; ++c) { g_cnt.find_candidates += info[2 * c]; g_cnt.find_steps += info[2 * c + 1]; }
    }
    g_cnt.find_s += now_s() - t_find0;
    // 2. jobs: from every start to the next one
    pg.t_dec0 = now_s();
    pg.launches.emplace_back();
    {
      Launch& L = pg.launches[0];
      L.jobs.push_back(Job{next_bit, 0, 0, 0, first ? (uint32_t)(F_HEADER | F_MEMBER_START) : 0u, 0});
      for (uint64_t s : starts)
        if (s != ~0ull) L.jobs.push_back(Job{s, 0, 0, 0, 0, 0});
      std::vector<uint64_t> caps(L.jobs.size());
      for (size_t i = 0; i < L.jobs.size(); ++i) {
        L.jobs[i].stop_bit = i + 1 < L.jobs.size() ? L.jobs[i + 1].start_bit : limit_bit;
        const uint64_t span_end = i + 1 < L.jobs.size() ? L.jobs[i + 1].start_bit : scan_end;
        caps[i] = ((span_end - L.jobs[i].start_bit) / 8 + 1) * g_cfg.ratio + 4096;
        // (the stage's last job runs on to the first block boundary BEHIND the limit: room for a block more, or every stage has a
        // job that counts on and is decoded a second time, alone on the device)
        if (i + 1 == L.jobs.size() && limit_bit != ~0ull) caps[i] += 2 * g_cfg.chunk_bytes * g_cfg.ratio;
      }
      events_.clear();
      MG_TRY(launch_jobs(&L, caps, final, avail));
    }
    return MG_OK;
  }
  int gz_finish(uint64_t headroom, DevBuf* text, uint64_t* ntext, bool* finished) {
    *ntext = 0;
    *finished = false;
    pg.active = false;
    if (pg.nothing) return MG_OK;
    const bool final = pg.final;
    const uint64_t avail = pg.avail, limit_bit = pg.limit_bit;
    const double t_begin = pg.t_begin, t_dec0 = pg.t_dec0;
    std::deque<Launch>& launches = pg.launches;
    MG_TRY(collect_jobs(&launches[0], &events_, 0));
    // jobs that needed more room than was reserved: all of them again in ONE launch (they counted what they need)
    std::vector<uint32_t> again_of(launches[0].jobs.size(), ~0u);
    {
      std::vector<uint64_t> caps;
      std::vector<Job> again;
      for (size_t j = 0; j < launches[0].jobs.size(); ++j)
        if (launches[0].res[j].overflow) {
          again_of[j] = (uint32_t)again.size();
          again.push_back(launches[0].jobs[j]);
          caps.push_back(launches[0].res[j].out_count + 64);
        }
      if (!again.empty()) {
        g_cnt.redone += again.size();
        launches.emplace_back();
        launches.back().jobs = again;
        MG_TRY(run_launch(&launches.back(), caps, final, avail, &events_, (uint32_t)launches.size() - 1));
      }
    }
    const bool trace = dbg("inflate_trace") != 0;
    if (trace) std::fprintf(stderr, "[inflate] stage at bit %llu: %zu jobs, %zu decoded again for room\n", (unsigned long long)next_bit, launches[0].jobs.size(), launches.size() > 1 ? launches[1].jobs.size() : (size_t)0);
    // 3. the chain: every job must have ended where the next one started
    struct Link { uint32_t launch, job; };
    std::vector<Link> chain;
    bool stream_end = false;
    {
      uint32_t li = 0, ji = 0;  // the job the chain stands at
      size_t main_next = 1;     // the first job of launch 0 that has not been passed
      for (;;) {
        Launch& L = launches[li];
        Result& r = L.res[ji];
        if (r.overflow && li == 0 && again_of[ji] != ~0u) {  // more symbols than were reserved: decoded again above
          li = 1;
          ji = again_of[ji];
          continue;
        }
        if (r.overflow) {
          launches.emplace_back();
          Launch& R = launches.back();
          R.jobs.push_back(launches[li].jobs[ji]);
          MG_TRY(run_launch(&R, {launches[li].res[ji].out_count + 64}, final, avail, &events_, (uint32_t)launches.size() - 1));
          g_cnt.redone += 1;
          li = (uint32_t)launches.size() - 1;
          ji = 0;
          continue;
        }
        if (r.status >= ST_ERR) return fail(MG_ERR_ARG, "%s (near compressed byte %llu)", status_text(r.status), (unsigned long long)(r.end_bit / 8));
        if (r.status == ST_NEED_MORE) {
          if (final) return fail(MG_ERR_ARG, "%s", status_text(ST_TRUNC));
          break;  // the stage ends in front of this job
        }
        chain.push_back(Link{li, ji});
        if (r.status == ST_END) { stream_end = true; break; }
        if (r.status != ST_STOP) return fail(MG_ERR_STATE, "inflate: job ended in state %u", r.status);
        const uint64_t end = r.end_bit;
        if (limit_bit != ~0ull && end >= limit_bit) break;  // the stage's last boundary
        while (main_next < launches[0].jobs.size() && launches[0].jobs[main_next].start_bit < end) ++main_next;  // (run over: false starts)
        if (main_next < launches[0].jobs.size() && launches[0].jobs[main_next].start_bit == end) {
          li = 0;
          ji = (uint32_t)main_next++;
          continue;
        }
        // a hole: nobody started where this job ended
        if (trace) std::fprintf(stderr, "[inflate] hole at bit %llu (the next start: %lld bits on)\n", (unsigned long long)end, main_next < launches[0].jobs.size() ? (long long)(launches[0].jobs[main_next].start_bit - end) : -1ll);
        launches.emplace_back();
        Launch& R = launches.back();
        const uint64_t stop = main_next < launches[0].jobs.size() ? launches[0].jobs[main_next].start_bit : limit_bit;
        const uint64_t span_end = stop == ~0ull ? avail * 8 : stop;
        R.jobs.push_back(Job{end, stop, 0, 0, 0, 0});
        MG_TRY(run_launch(&R, {((span_end > end ? span_end - end : 0) / 8 + 1) * g_cfg.ratio + 4096}, final, avail, &events_, (uint32_t)launches.size() - 1));
        g_cnt.redone += 1;
        li = (uint32_t)launches.size() - 1;
        ji = 0;
      }
    }
    g_cnt.decode_s += now_s() - t_dec0;
    if (chain.empty()) return MG_OK;  // (not final, and the first job already ran out of bytes)
    // 4. where every job's text goes
    const double t_res0 = now_s();
    const size_t nc = chain.size();
    std::vector<ChainJob> cj(nc);
    uint64_t total = 0, tiles = 0;
    for (size_t i = 0; i < nc; ++i) {
      Launch& L = launches[chain[i].launch];
      const Result& r = L.res[chain[i].job];
      cj[i].sym = L.sym.as<uint16_t>() + L.jobs[chain[i].job].out_off;
      cj[i].tail = L.tails.as<uint16_t>() + (uint64_t)chain[i].job * kWindow;
      cj[i].n = r.out_count;
      cj[i].t_off = headroom + total;
      cj[i].tile0 = tiles;
      total += r.out_count;
      tiles += r.out_count ? ((cj[i].t_off + r.out_count - (cj[i].t_off & ~15ull)) + 4095) / 4096 : 0;
    }
    MG_TRY(text->alloc(headroom + total + 64));
    DevBuf dcj, wins;
    MG_TRY(dcj.alloc(nc * sizeof(ChainJob)));
    MG_TRY(wins.alloc((uint64_t)(nc + 1) * kWindow));
    MG_HIP(hipMemcpyAsync(dcj.p, cj.data(), nc * sizeof(ChainJob), hipMemcpyHostToDevice, st));
    MG_HIP(hipMemcpyAsync(wins.p, win.p, kWindow, hipMemcpyDeviceToDevice, st));
    {
      // groups of K jobs: K steps in every group at once, then one step per group
      uint32_t K = 1;
      while ((uint64_t)K * K < nc) ++K;
      const uint32_t G = (uint32_t)((nc + K - 1) / K);
      DevBuf comp, tops;
      MG_TRY(comp.alloc((uint64_t)nc * kWindow * 2));
      MG_TRY(tops.alloc((uint64_t)(G + 1) * kWindow));
      MG_HIP(hipMemcpyAsync(tops.p, win.p, kWindow, hipMemcpyDeviceToDevice, st));
      static bool attr = false;
      if (!attr) {
        MG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_groups), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kWindow));
        attr = true;
      }
      ProfScope ps("k_window_chain", st);
      k_chain_groups<<<G, 1024, 4 * kWindow, st>>>(dcj.as<ChainJob>(), (uint32_t)nc, K, comp.as<uint16_t>());
      k_chain_tops<<<1, 1024, 2 * kWindow, st>>>(comp.as<uint16_t>(), (uint32_t)nc, K, G, tops.as<uint8_t>());
      k_chain_rows<<<(unsigned)(((uint64_t)nc * (kWindow / 16) + 255) / 256), 256, 0, st>>>(comp.as<uint16_t>(), (uint32_t)nc, K, tops.as<uint8_t>(), wins.as<uint8_t>());
    }
    if (tiles) {
      ProfScope ps("k_resolve_text", st);
      k_resolve_text<<<(unsigned)tiles, 256, 0, st>>>(dcj.as<ChainJob>(), (uint32_t)nc, wins.as<uint8_t>(), text->as<uint8_t>());
    }
    MG_HIP(hipMemcpyAsync(win.p, wins.as<uint8_t>() + (uint64_t)nc * kWindow, kWindow, hipMemcpyDeviceToDevice, st));
    // 5. the members' CRC-32 and ISIZE: the text between member ends, in segments of 256 KB
    std::vector<std::pair<uint64_t, Event>> ends;  // (position in the stage's text, trailer)
    {
      std::map<uint64_t, size_t> where;
      for (size_t i = 0; i < nc; ++i) where[(uint64_t)chain[i].launch << 32 | chain[i].job] = i;
      for (const Event& e : events_) {  // (member ends seen by jobs that are not on the chain are nobody's)
        const auto it = where.find((uint64_t)e.pad << 32 | e.job);
        if (it != where.end()) ends.emplace_back(cj[it->second].t_off - headroom + e.out_pos, e);
      }
    }
    std::stable_sort(ends.begin(), ends.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    const uint64_t kSeg = 256u << 10;
    std::vector<CrcSeg> segs;
    std::vector<size_t> seg_end_of;  // for every member end: segments before it
    {
      uint64_t at = 0;
      size_t ei = 0;
      while (at < total || ei < ends.size()) {
        const uint64_t until = ei < ends.size() ? ends[ei].first : total;
        while (at < until) {
          const uint64_t len = std::min(kSeg, until - at);
          segs.push_back(CrcSeg{headroom + at, len});
          at += len;
        }
        if (ei < ends.size()) { seg_end_of.push_back(segs.size()); ++ei; }
        else break;
      }
    }
    std::vector<uint32_t> crcs;
    MG_TRY(crc_of(text->as<uint8_t>(), segs, &crcs));
    {
      size_t si = 0;
      for (size_t ei = 0; ei <= ends.size(); ++ei) {
        const size_t upto = ei < ends.size() ? seg_end_of[ei] : segs.size();
        for (; si < upto; ++si) {
          mcrc = mlen ? crc_multmodp(x8n(segs[si].len), mcrc) ^ crcs[si] : crcs[si];
          mlen += segs[si].len;
        }
        if (ei < ends.size()) {
          const Event& e = ends[ei].second;
          if ((mlen ? mcrc : 0u) != e.crc) return fail(MG_ERR_ARG, "corrupt gzip data: CRC mismatch in the member that ends at byte %llu of the text", (unsigned long long)(text_base_ + ends[ei].first));
          if ((uint32_t)mlen != e.isize) return fail(MG_ERR_ARG, "corrupt gzip data: length mismatch in the member that ends at byte %llu of the text", (unsigned long long)(text_base_ + ends[ei].first));
          mcrc = 0;
          mlen = 0;
        }
      }
    }
    g_cnt.resolve_s += now_s() - t_res0;
    const Launch& LL = launches[chain.back().launch];
    seen_jobs += chain.size();
    seen_bytes += (LL.res[chain.back().job].end_bit - next_bit) / 8;
    next_bit = LL.res[chain.back().job].end_bit;
    first = false;
    text_base_ += total;
    *ntext = total;
    if (stream_end) {
      if (mlen) return fail(MG_ERR_ARG, "%s", status_text(ST_TRUNC));
      *finished = true;
    }
    g_cnt.stages += 1;
    g_cnt.stage_s += now_s() - t_begin;
    launches.clear();
    return MG_OK;
  }

  // the next piece of text; *finished: nothing follows
  // The next piece of text in two halves: begin() returns with the stage's decoder running on the inflater's stream (the caller's
  // work on the library's main stream — parsing and hashing the previous stage's text — runs beside it); finish() -> the text.
  // *finished: nothing follows.
  bool pending = false;
  int begin() {
    if (done || pending) return MG_OK;
    pending = true;
    if (bgzf) return bz_begin(768ull << 20);
    // a stage needs its own compressed bytes and a little more (its last job reads on to the end of its block)
    const uint64_t avail = std::min<uint64_t>(n, next_bit / 8 + stage_size() + (8ull << 20));
    return gz_begin(avail, avail == n);
  }
  int finish(DevBuf* text, uint64_t* ntext, bool* finished) {
    if (done) { *ntext = 0; *finished = true; return MG_OK; }
    if (!pending) MG_TRY(begin());
    pending = false;
    int rc;
    if (bgzf) {
      rc = bz_finish(text, ntext, finished);
    } else {
      const bool was_final = pg.final;
      rc = gz_finish(headroom, text, ntext, finished);
      if (rc == MG_OK && *ntext == 0 && !*finished && !was_final) {  // (a block longer than the margin: with every byte there is)
        rc = gz_begin(n, true);
        if (rc == MG_OK) rc = gz_finish(headroom, text, ntext, finished);
      }
    }
    if (rc == MG_OK && *finished) done = true;
    return rc;
  }
  ~DevInflater() {
    if (st) (void)hipStreamSynchronize(st);  // (a stage may still be running when an error ends the pipeline: before its buffers go)
    ctx().inf_side = false;
  }

  std::vector<Event> events_;
  uint64_t text_base_ = 0;
};

// the pipeline of mg_stream.hip's entry points for a .gz file: every stage's text, the unfinished last record carried in front of
// the next stage's
int inflate_file_pipeline(int fd, uint64_t fsize, const std::function<int(const uint8_t*, uint64_t, bool, uint64_t*)>& consume,
                          bool* started) {
  if (started) *started = false;
  if (fsize == 0) return fail(MG_ERR_ARG, "empty file: not in gzip format");
  void* map = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
  if (map == MAP_FAILED) return fail(MG_ERR_ARG, "cannot map the file: %s", strerror(errno));
  (void)madvise(map, fsize, MADV_SEQUENTIAL);
  int rc = MG_OK;
  {
    DevInflater inf;
    const uint64_t headroom = 4ull << 20;
    rc = inf.open(static_cast<const uint8_t*>(map), fsize, 4, headroom);
    DevBuf prev;
    const uint8_t* carry_src = nullptr;
    uint64_t carry = 0;
    hipStream_t st = ctx().stream;
    if (rc == MG_OK) rc = inf.begin();
    while (rc == MG_OK) {
      DevBuf text;
      uint64_t nt = 0;
      bool fin = false;
      rc = inf.finish(&text, &nt, &fin);
      if (rc != MG_OK) break;
      if (!fin) {  // the next stage's decoder starts now and runs beside the consumer of this stage's text
        rc = inf.begin();
        if (rc != MG_OK) break;
      }
      if (carry) {
        if (hipMemcpyAsync(text.as<uint8_t>() + headroom - carry, carry_src, carry, hipMemcpyDeviceToDevice, st) != hipSuccess) { rc = fail(MG_ERR_HIP, "carry copy failed"); break; }
      }
      const uint8_t* d = text.as<uint8_t>() + headroom - carry;
      const uint64_t nbytes = carry + nt;
      uint64_t consumed = 0;
      if (started) *started = true;
      rc = consume(d, nbytes, fin, &consumed);
      if (rc != MG_OK || fin) break;
      if (consumed > nbytes) consumed = nbytes;
      carry = nbytes - consumed;
      if (carry > headroom) { rc = fail(MG_ERR_CAPACITY, "a record of more than %llu bytes does not fit the streaming pieces", (unsigned long long)headroom); break; }
      carry_src = d + consumed;
      prev = std::move(text);  // (kept until the carry has been copied out of it)
    }
    (void)hipStreamSynchronize(st);
    inf.up.finish();
  }
  munmap(map, fsize);
  return rc;
}

}  // namespace mg

using namespace mg;

struct mg_inflated {
  std::vector<mg::DevBuf> parts;
  std::vector<uint64_t> sizes;
  uint64_t headroom = 0, total = 0;
};

extern "C" {

int mg_inflate_config(int64_t chunk_bytes, int64_t stage_bytes, int ratio, int on, int64_t lane_jobs) {
  if (lane_jobs >= 0) g_cfg.lane_jobs = lane_jobs > 0xffffffffll ? 0xffffffffu : (uint32_t)lane_jobs;
  if (chunk_bytes > 0) {
    if (chunk_bytes < 1024) return fail(MG_ERR_ARG, "inflate chunks of at least 1024 bytes");
    g_cfg.chunk_bytes = (uint64_t)chunk_bytes;
  }
  if (stage_bytes > 0) {
    if (stage_bytes < 4096) return fail(MG_ERR_ARG, "inflate stages of at least 4096 bytes");
    g_cfg.stage_bytes = (uint64_t)stage_bytes;
  } else if (stage_bytes < 0) {
    g_cfg.stage_bytes = 0;  // sized by the device
  }
  if (ratio > 0) g_cfg.ratio = (uint32_t)ratio;
  if (on >= 0) g_cfg.on = on;
  return MG_OK;
}

int mg_inflate_stats(mg_inflate_counters* out, int reset) {
  if (out) *out = g_cnt;
  if (reset) g_cnt = mg_inflate_counters();
  return MG_OK;
}

int mg_inflate_dev(const uint8_t* comp, uint64_t ncomp, mg_inflated** out) {
  MG_REQUIRE_READY();
  if (!comp || !out) return fail(MG_ERR_ARG, "null argument");
  *out = nullptr;
  std::unique_ptr<mg_inflated> res(new mg_inflated());
  DevInflater inf;
  MG_TRY(inf.open(comp, ncomp, 4, 0));
  for (;;) {
    mg::DevBuf text;
    uint64_t nt = 0;
    bool fin = false;
    MG_TRY(inf.finish(&text, &nt, &fin));
    res->total += nt;
    res->sizes.push_back(nt);
    res->parts.push_back(std::move(text));
    if (fin) break;
  }
  MG_HIP(hipStreamSynchronize(ctx().stream));
  *out = res.release();
  return MG_OK;
}

uint64_t mg_inflated_bytes(const mg_inflated* t) { return t ? t->total : 0; }

int mg_inflated_download(const mg_inflated* t, uint8_t* dst) {
  MG_REQUIRE_READY();
  if (!t || (!dst && t->total)) return fail(MG_ERR_ARG, "null argument");
  uint64_t at = 0;
  for (size_t i = 0; i < t->parts.size(); ++i) {
    if (t->sizes[i]) MG_HIP(hipMemcpyAsync(dst + at, t->parts[i].as<uint8_t>(), t->sizes[i], hipMemcpyDeviceToHost, ctx().stream));
    at += t->sizes[i];
  }
  MG_HIP(hipStreamSynchronize(ctx().stream));
  return MG_OK;
}

void mg_inflated_free(mg_inflated* t) { delete t; }

}  // extern "C"
