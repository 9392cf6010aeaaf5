// Persistent denominator recursion, second form: no capacity cliff, and half of the table copy under the arcs (gfx950).
//
// chain_den_persist.hip keeps ALL arcs of a recursion in registers and the WHOLE state vector in LDS, which bounds the
// graphs it takes (~1.05 M arc slots, ~36 k states) -- one arc more and a call fell to the launch-per-frame kernels, 2.2x
// slower.  Same recursion per XCD, same exchange through the XCD's L2 here, but a frame's row sums are built from PASSES:
//  * the state vector is cut into table chunks (chain_internal.h: HostPersist2).  A vector that fits LDS has two, back to
//    back: chunk 0 is copied (LDS-DMA) when the frame's words are valid; pass A -- 32 register slots per thread, arcs that
//    gather from chunk 0 -- runs on it while the copy of chunk 1 is issued FROM INSIDE the pass, one instruction behind every
//    group of 8 gathers / multiply-adds (a wave that issues copies back to back stalls at the issue: the DMA queue is
//    shallow); pass B -- the other 32 slots, gathering from both chunks -- follows when chunk 1 is there.  A longer vector
//    goes through two half-size LDS buffers, chunk c+2 copied into the buffer chunk c has just left;
//  * arcs that do not fit the 2 x 32 register slots of a thread (and all arcs gathering from chunks >= 2) are STREAMED: read
//    again in every frame from the XCD's L2 in pieces of 8 slots per thread, the next piece on its way while one is summed.
//    A graph a little too large costs a few pieces per frame, not a fall to another kernel family;
//  * every pass sums complete rows in a lane / finishes rows crossing lanes by a segmented wave scan / leaves the wave's
//    open tail as a carry, exactly as den_persist_kernel does for its single list; pass A and pass B store into their own
//    compact LDS row arrays (one plain store per row that has an arc in the list), streamed segments add into a row-indexed
//    one, the row epilogue adds up the arrays and the carries of all segments.
// Everything else -- teams by arrival order per XCD, the task queue, NaN-sentinel words, one exchange per frame in both
// directions, history stores after the words other workgroups wait for, the 1 s poll timeout -- is chain_den_persist.hip's
// (den_persist_dev.h).  Replaces the same DenominatorComputation (reference ops/ops.py:265, bin/train_chain.py:202).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>
#include <vector>

#include "chain_num.h"
#include "den_persist_dev.h"

namespace pk2 {

#if PK2_DP_LOADMODE == 0
#define PK2_DMA_SC " sc1"
#else
#define PK2_DMA_SC ""
#endif

struct DenPersist2Params {
  DenParams d;
  DevPersist2 fwd, bwd;
  const float* xv;        // [G][Tmax][V]; with `xgather`: exp(logits) [G][Tmax][P], gathered by pdf (round 4)
  const int32_t* vpdf;    // [V] pdf of a virtual state (-1: x = 1)
  const int32_t* loop_pdf;// [S] pdf of a state's peeled self-loop
  int xgather;
  int rowarrays;          // LDS arrays of `cap` floats (den_persist.h)
  float* ring;            // [8 * kMaxTeams][2][rpad]
  float* pring;           // [8 * kMaxTeams][3][kPR][kPWords]
  int rpad;               // floats per ring slot
  int tfloats;            // LDS table floats
  int cap;                // LDS row buffers
  int ntasks;                   // recursions first (longest first), then the numerators (task_dir 2)
  NumParams np;                 // the minibatch's numerator forward-backward, if it rides in this launch
  int fwd_stream, bwd_stream;   // the ordering has streamed pieces (or more than two table chunks)
  int pspt;                     // 2 or kPSPT: epilogue entries per thread
  short task_seq[kMaxTasks];
  unsigned char task_dir[kMaxTasks];
};
typedef __attribute__((address_space(4))) const DenPersist2Params CParams2;
typedef __attribute__((address_space(4))) const DevPersist2 CDev2;
__device__ __forceinline__ CParams2* uni(CParams2* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (CParams2*)(((unsigned long long)hi << 32) | lo);
}

extern __shared__ __attribute__((aligned(16))) float den_persist2_smem[];
struct Lds2 {
  float* table;    // [tfloats] the chunks of the frame's gather table
  float* accA;     // [cap]  row sums of pass A, by compact row of list 0
  float* accB;     // [cap]  row sums of pass B, by compact row of list 1
  float* accS;     // [cap]  row sums of the streamed segments, by rank-local row
  float* xown;     // [cap]  forward: x[t, own rows]; backward: x of own virtual states
  float* leak;     // [cap]
  float* aux;      // [cap]  backward: weight of an own virtual state in lU; forward: the finished rows (history stores)
  short* mapA;     // [cap]  compact row of a rank-local row in list 0 / list 1 (-1: the row has no slot there)
  short* mapB;     // [cap]
  short* pdfv;     // [cap]  pdf of an own row (forward) / own virtual state (backward): x gathered from the frame's exp row
  short* pdfl;     // [cap]  pdf of an own state's peeled self-loop
  float* red;      // [2 * kPW]
  float* tot;      // [4]
  float* wcarry;   // [kSegs * kPW] open tail of wave w in segment s
  int* wcrow;      // [kSegs * kPW] the rank-local row it belongs to (-1: none)
  int* abort;      // + rank, team, xcd, task; [8] = ranks whose words have been seen valid in the poll under way
  uint32_t* need;  // [kP2NeedRows] ranks whose slices a 1 KB row of chunk 0 holds
};
__device__ __forceinline__ Lds2 carve_lds2(int tfloats, int cap, int arrays) {
  Lds2 L;
  L.table = den_persist2_smem; L.accA = L.table + tfloats; L.accB = L.accA + cap; L.accS = L.accB + cap; L.xown = L.accS + cap;
  L.leak = L.xown + cap; L.aux = L.leak + cap;
  L.mapA = reinterpret_cast<short*>(L.aux + cap); L.mapB = L.mapA + cap;
  L.pdfv = L.mapB + cap; L.pdfl = L.pdfv + cap;
  L.red = L.aux + (arrays - 5) * cap; L.tot = L.red + 2 * kPW; L.wcarry = L.tot + 4;      // (without the pdfs: 7 arrays)
  L.wcrow = reinterpret_cast<int*>(L.wcarry + kSegs * kPW); L.abort = L.wcrow + kSegs * kPW;
  L.need = reinterpret_cast<uint32_t*>(L.abort + 16);
  return L;
}

// Workgroup barrier that orders LDS accesses only: the LDS-DMA instructions a wave has in flight stay in flight.
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Loads / stores of data touched once per call (x of the frames, the alpha / beta history): non-temporal in the kernels
// WITHOUT streamed pieces (round 5; -DPK2_DP2_NT=0 for plain ones).  Measured (profiles/r05_den_stream.txt): the bench graph
// 7.24 -> 7.04 us per frame -- the once-only lines no longer displace the state vectors the 32 ranks copy in every frame
// from the XCD's L2; the streaming kernels got SLOWER with them (S = 30 k, 1.5 M arcs: 11.2 -> 12.7 us; the stores then go
// out to memory at once, over the fabric the pieces come in by), so they keep plain accesses.
#ifndef PK2_DP2_NT
#define PK2_DP2_NT 1
#endif
// The history stores of a frame (alpha / beta-hat for the parallel passes: nobody in this kernel waits for them) follow the
// publication of the rank's slice -- ahead of the next frame's poll, whose first answer (vmcnt counts in order) comes back
// behind their acknowledgements.  Round 5, measured and left off (-DPK2_DP2_LATE_HISTORY=1): the stores leaving from the
// NEXT frame's staging point instead (behind the wait for table chunk 1, where every older memory operation has completed)
// -- 7.16 us per frame against 7.07 on the bench graph, 5.10 against 5.03 at S = 10 k, two rounds each on one box
// (profiles/r05_den_stream.txt): the poll's 0.6 us behind the stores is time the rank would spend waiting for the slowest
// rank's words anyway, and the stores then compete with the x prefetch and pass B's row-end traffic.
#ifndef PK2_DP2_LATE_HISTORY
#define PK2_DP2_LATE_HISTORY 0
#endif
#ifndef PK2_DP2_LATE_STREAM
#define PK2_DP2_LATE_STREAM 1          // ... in the kernels with streamed pieces too
#endif
template <bool STREAM> constexpr bool kLateHistory = PK2_DP2_LATE_HISTORY && (!STREAM || PK2_DP2_LATE_STREAM);
// (round 6, VERDICT r5 #2c: loads and stores switchable apart -- -DPK2_DP2_NT_LOAD=0 / -DPK2_DP2_NT_STORE=0)
#ifndef PK2_DP2_NT_LOAD
#define PK2_DP2_NT_LOAD PK2_DP2_NT
#endif
#ifndef PK2_DP2_NT_STORE
#define PK2_DP2_NT_STORE PK2_DP2_NT
#endif
template <bool NT>
__device__ __forceinline__ float once_load(cgfloat* p) {
  if constexpr (NT && PK2_DP2_NT_LOAD) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
__device__ __forceinline__ void once_store(gfloat* p, float v) {
  if constexpr (NT && PK2_DP2_NT_STORE) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// The streaming kernels sit at the 256-register limit, and whatever the compiler hoists out of the frame loop (per-thread
// 64-bit addresses, LDS addresses of the thread's map entries) it spills -- and reloads with s_waitcnt vmcnt(0), behind the
// streamed pieces in flight.  An index made opaque per frame is recomputed instead (one or two VALU instructions).
template <bool ON>
__device__ __forceinline__ int per_frame(int r) {
  if constexpr (ON) asm volatile("" : "+v"(r));
  return r;
}
// x[ubase + idx] with a uniform base and a 32-bit per-thread index: SGPR base + VGPR offset, no 64-bit per-thread address.
__device__ __forceinline__ float ld_uniform_base(cgfloat* ubase, int idx) {
  typedef __attribute__((address_space(1))) const char gchar;
  return *(cgfloat*)((gchar*)ubase + (uint32_t)idx * 4u);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate has to be a constant).  As the BUILTIN, not as inline
// assembly (round 5): the compiler's own wait-count pass reads an s_waitcnt instruction it finds and knows afterwards that
// its earlier loads have landed; behind an opaque asm it kept waiting for registers loaded frames ago with vmcnt(few) --
// which, the counter being in order, also waited for the LDS-DMA copies issued in between (a streamed piece consumed during
// the copy of chunk 1 stalled until that copy was nearly complete).
__device__ __forceinline__ constexpr int vmcnt_imm(int k) { return (k & 15) | ((k >> 4) << 14) | 0x70 | 0xF00; }     // gfx9: expcnt, lgkmcnt left alone
__device__ __forceinline__ void wait_vm(int n) {
  asm volatile("" ::: "memory");
  switch (__builtin_amdgcn_readfirstlane(n)) {
#define PK2_VM_CASE(k) case k: __builtin_amdgcn_s_waitcnt(vmcnt_imm(k)); break;
    PK2_VM_CASE(1) PK2_VM_CASE(2) PK2_VM_CASE(3) PK2_VM_CASE(4) PK2_VM_CASE(5) PK2_VM_CASE(6) PK2_VM_CASE(7) PK2_VM_CASE(8)
    PK2_VM_CASE(9) PK2_VM_CASE(10) PK2_VM_CASE(11) PK2_VM_CASE(12) PK2_VM_CASE(13) PK2_VM_CASE(14) PK2_VM_CASE(15) PK2_VM_CASE(16)
    PK2_VM_CASE(17) PK2_VM_CASE(18) PK2_VM_CASE(19) PK2_VM_CASE(20)
#undef PK2_VM_CASE
    default: __builtin_amdgcn_s_waitcnt(vmcnt_imm(0)); break;
  }
  asm volatile("" ::: "memory");
}

// LDS-DMA of table chunk c of the vector at `src` into its LDS buffer: 1 KB rows dealt to the waves round robin; every wave
// issues the SAME number of instructions (a wave without a row of its own copies the last row again), which is returned: the
// count a later s_waitcnt may leave in flight.
// Round 5, measured and left off: the ranks of a team copy the SAME vector at the same time, starting at the same row --
// would they be faster if rank r started `rank * rows / kPR` rows into the chunk and wrapped around, so that the 32 CUs ask
// 32 different L2 channels at any moment (-DPK2_DP2_ROTATE=1)?  No: 7.30 us per frame against 7.22 (same job, two runs
// each, profiles/r05_den_ab.txt).  Many CUs reading one line at about the same time is what an XCD's L2 serves best; the
// copy rate of ~55 KB/us per CU is the CU's own limit (1 KB LDS-DMA instructions in flight x L2 latency), not a channel's.
#ifndef PK2_DP2_ROTATE
#define PK2_DP2_ROTATE 0
#endif
__device__ __forceinline__ int dma_rot(int rank, int rows) { return PK2_DP2_ROTATE ? (rank * rows) / kPR : 0; }
__device__ __forceinline__ int dma_chunk(cgfloat* src, CDev2& o, int c, float* table, int rank) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = o.cbeg[c], e = o.cbeg[c + 1];
  const int rows = (e - b + 255) >> 8;
  const int n = (rows + kPW - 1) / kPW;
  const int e4 = (e + 3) & ~3;
  const int rot = dma_rot(rank, rows);
  float* dst = table + o.lds_off[c];
  for (int k = 0; k < n; ++k) {
    int row = w + k * kPW;
    row = row < rows ? row : rows - 1;
    row += rot; row = row >= rows ? row - rows : row;
    const int off = row << 8;
    if (b + off + lane * 4 < e4) dma256(src + b + off + lane * 4, dst + off);
  }
  return n;
}
// The copy of chunk 1, issued from INSIDE pass A.  An LDS-DMA instruction does not retire into a deep queue: a CU moves
// ~60 KB/us and a wave that issues copies faster than that stalls at the issue (measured: 16 back-to-back instructions per
// wave take 2.1 us to issue).  So "start both chunks, compute on chunk 0 meanwhile" does not work from one instruction stream
// -- the stream is stuck issuing until chunk 1 has all but landed.  Instead one copy instruction of chunk 1 is placed after
// every group of 8 gathers and after every group of 8 multiply-adds of pass A (8 places: 64 rows = 64 KB per workgroup,
// about what the pass takes to run): each finds room in the queue, and the pass and the copy finish together.
// LDS byte address of a pointer into dynamic shared memory, and plain LDS accesses through such addresses (ds_read_b32 /
// ds_write_b32 with the address register as it stands: no base to add).
typedef __attribute__((address_space(3))) float lfloat;
__device__ __forceinline__ uint32_t lds_addr(const float* p) { return (uint32_t)(uintptr_t)(const lfloat*)p; }
__device__ __forceinline__ float lds_load(uint32_t a) { return *(const lfloat*)(uintptr_t)a; }
__device__ __forceinline__ void lds_store(uint32_t a, float v) { *(lfloat*)(uintptr_t)a = v; }

// The places of one task never change: which row a wave copies at place j, and which of its lanes lie inside the vector,
// are formed once per task (DmaPlan, scalar registers); a frame adds its source address.
struct DmaPlan {
  uint64_t mask[8];  // lanes of the wave's row at place j that lie inside the vector (0: the wave has no row there)
  uint32_t off4[8];  // byte offset of that row from the first row of chunk 1
  uint32_t dst;      // LDS byte address of row 0 of chunk 1
  int rows, efl, w;  // rows of the chunk; floats from its first to the (granule-rounded) end of the vector; wave
  int rot;           // first row of this rank (dma_rot)
  uint32_t lane16;   // lane * 16: the per-lane part of the address
  __device__ __forceinline__ void place(int j, uint64_t* mask_out, uint32_t* off4_out) const {
    int row = w + j * kPW;                    // (every wave copies ceil(rows / kPW) rows, the last one again if it has
    row = row < rows ? row : rows - 1;        //  none of its own -- the counts stay equal across the waves)
    row += rot; row = row >= rows ? row - rows : row;
    const int off = row << 8;
    int n = (efl - off + 3) >> 2;             // lanes (16 bytes each) inside the vector
    n = j * kPW < rows ? n : 0;
    n = n < 0 ? 0 : n;
    *mask_out = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    *off4_out = (uint32_t)off * 4u;
  }
  __device__ __forceinline__ void init(CDev2& o, const float* table, int rank) {
    const int b1 = o.cbeg[1], e1 = o.cbeg[2];
    dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(table + o.lds_off[1]));
    rows = (e1 - b1 + 255) >> 8; efl = ((e1 + 3) & ~3) - b1; w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    lane16 = (threadIdx.x & 63u) * 16u;
    rot = dma_rot(rank, rows);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      place(j, &mask[j], &off4[j]);
      // (opaque: scalar registers to keep, not expressions to evaluate again in every frame)
      asm volatile("" : "+s"(mask[j]), "+s"(off4[j]));
    }
  }
};
__device__ __forceinline__ void dma_issue(uint64_t mask, uint32_t m0v, uint32_t lane16, uint64_t g) {
  // (exec is saved and restored instead of being assumed all ones.  m0 cannot be declared: it is a reserved
  // register to LLVM, which answers a clobber entry with "reserved registers on the clobber list may not be preserved" and
  // honours nothing; the compiler's own m0 uses in this kernel are the global_load_lds builtins, each of which sets m0
  // immediately before the instruction (checked in the ISA), and -DPK2_DP2_PASS_ASM=0 keeps the builtin-only build as the
  // parity A/B of this statement.)
  uint64_t saved;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %3, %4" PK2_DMA_SC "\n\ts_mov_b64 exec, %0"
               : "=&s"(saved) : "s"(mask), "s"(m0v), "v"(lane16), "s"(g) : "memory");
}
struct Chunk1Dma {
  uint64_t src;      // byte address of row 0 of chunk 1 in global memory (wave-uniform)
  const DmaPlan* pl;
  // Round 5: branch-free, and no VALU work.  The copy used to sit in two nested conditional blocks (wave-uniform "this wave
  // still has a row", per-lane "inside the vector"); at the join behind every one of them the compiler waited for ALL
  // outstanding LDS reads (s_waitcnt lgkmcnt(0): 8 times per pass A), so the gathers issued ahead for the next group of 8
  // slots were waited for on the spot and the double buffer hid nothing; and each place cost 9 VALU instructions of address
  // and predicate arithmetic per wave and frame.  Here the row, its LDS and global addresses and the lane mask are scalar
  // values (the lanes inside the vector are the first n of the wave), the predicate is the exec mask around one
  // instruction inside a single asm statement -- no control flow -- and the global address is scalar base + lane * 16.
  // (The compiler does not know of the extra vmcnt event: its own vmcnt waits only become stricter; the frame waits with
  // vmcnt(0) for the copy.  exec is all ones here: every thread of the workgroup runs the passes.)
  __device__ __forceinline__ void operator()(int j) const {
    if (j < 8) { dma_issue(pl->mask[j], pl->dst + pl->off4[j], pl->lane16, src + pl->off4[j]); return; }
    uint64_t mask; uint32_t off4;             // (a chunk 1 of more than 64 rows: the places behind the pass)
    pl->place(j, &mask, &off4);
    dma_issue(mask, pl->dst + off4, pl->lane16, src + off4);
  }
};
// Round 5, measured and left off (-DPK2_DP2_POLLCOPY=1): the copy of chunk 0 FOLLOWING the poll.  The frame waits until the
// words of all 32 ranks are valid and only then starts copying the vector -- 0.95 us of exposed copy behind ~0.7 us of poll,
// although a rank's word says that ITS slice is in L2.  Here wave 0 publishes, in an LDS word, the ranks it has seen valid so
// far, and waves 1..7 copy the 1 KB rows of chunk 0 (dealt to them round robin) as soon as the ranks whose slices a row holds
// (L.need, formed once per task from the ranks' slice bounds) are all there.  Parity-green -- and no faster: 7.27 us per frame
// against 7.20 (same job, two runs each; a first version whose copy loop read its need masks from LDS row by row: 8.2).
// The 0.7 us of the poll are the PATH of a poll (own stores acknowledged, one L2 round trip), not waiting for a slow rank:
// the ranks of a team arrive within ~0.1 us of each other, so there is nothing for the copy to hide under.
template <typename SLICE>
__device__ __forceinline__ void need_masks(const Lds2& L, CDev2& o, SLICE slice_begin) {      // slice_begin(r): first table index rank r publishes
  const int b = o.cbeg[0], e = o.cbeg[1];
  const int rows = (e - b + 255) >> 8;
  for (int j = threadIdx.x; j < rows && j < kP2NeedRows; j += kPT) {
    const int lo = b + (j << 8), hi = min(lo + 256, e);
    uint32_t m = 0;
    for (int r = 0; r < kPR; ++r)
      if (slice_begin(r) < hi && slice_begin(r + 1) > lo) m |= 1u << r;
    L.need[j] = m;
  }
  if (threadIdx.x == 0) L.abort[8] = 0;
}
// Wave 0: poll_words' loop, publishing its progress; waves 1..kPW-1: the copies.  Ends like poll_words (L.tot, L.abort) with
// every row of chunk 0 issued; the caller's barrier and wait_vm(0) follow.  Returns false when chunk 0 has more rows than
// L.need holds (the caller copies it the old way).
__device__ __forceinline__ bool poll_and_copy0(cgfloat* ps, int nwords, Spin& spin, const Lds2& L, cgfloat* src, CDev2& o) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = o.cbeg[0], e = o.cbeg[1];
  const int rows = (e - b + 255) >> 8;
  volatile uint32_t* prog = reinterpret_cast<volatile uint32_t*>(L.abort + 8);
  if (rows > kP2NeedRows) { poll_words(ps, 0, nwords, spin, L); return false; }
  if (w == 0) {
    const bool mine = (lane >> 5) < nwords;
    cgfloat* wsrc = ps + (lane & (kPR - 1)) * kPWords + (lane >> 5);
    float pv = mine ? ld_agent(wsrc) : 0.f;
    bool ok = true;
    for (;;) {
      const unsigned long long bad = __ballot(mine && is_sentinel(pv));
      // a rank is there when all its words are (backward: two)
      const uint32_t have = ~((uint32_t)bad | (uint32_t)(bad >> 32));
      if (lane == 0) *prog = have;
      if (bad == 0ull) break;
      if (spin.expired()) { ok = false; break; }
      if (mine && is_sentinel(pv)) pv = ld_agent(wsrc);
    }
    if (!ok) { pv = 0.f; if (lane == 0) { *L.abort = 1; *prog = 0xFFFFFFFFu; } }
    const float a = wave_sum_dpp(lane < kPR ? pv : 0.f), bsum = wave_sum_dpp(lane < kPR ? 0.f : pv);
    if (lane == 0) { L.tot[0] = a; L.tot[1] = bsum; }
  } else {
    // this wave's rows: w-1, w-1 + (kPW-1), ...: lane k looks after row k of them (its need mask in a register for the task
    // would be better still; an LDS read per poll sweep is what it costs here), one ballot says which are ready, and the
    // ready ones are issued by scalar code (dma_issue: exec mask + m0 + scalar base, as the in-pass copy)
    const int w1 = __builtin_amdgcn_readfirstlane(w - 1);        // (wave-uniform by construction: scalar registers below)
    const int nmine = (rows - w1 + (kPW - 2)) / (kPW - 1);
    const int myrow = w1 + lane * (kPW - 1);
    const uint32_t myneed = lane < nmine ? L.need[myrow] : 0xFFFFFFFFu;
    const int efl = ((e + 3) & ~3) - b;                 // floats from the chunk's first to the granule-rounded end
    const uint32_t dst0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(L.table + o.lds_off[0]));
    const uint64_t sb = (uint64_t)(src + b);
    const uint64_t src0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sb >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sb);
    const uint32_t lane16 = (uint32_t)lane * 16u;
    uint64_t pend = nmine >= 64 ? ~0ull : ((1ull << (nmine > 0 ? nmine : 0)) - 1ull);
    while (pend) {
      const uint32_t have = *prog;
      uint64_t ready = __ballot((myneed & ~have) == 0u) & pend;
      pend &= ~ready;
      while (ready) {
        const int k = __builtin_ctzll(ready);
        ready &= ready - 1;
        const int off = (w1 + k * (kPW - 1)) << 8;
        int nl = (efl - off + 3) >> 2;                   // lanes (16 bytes each) inside the vector
        nl = nl < 0 ? 0 : nl;
        const uint64_t mask = nl >= 64 ? ~0ull : ((1ull << nl) - 1ull);
        dma_issue(mask, dst0 + (uint32_t)off * 4u, lane16, src0 + (uint64_t)(uint32_t)off * 4u);
      }
    }
  }
  return true;
}

struct NoDma { __device__ __forceinline__ void operator()(int) const {} };

// Row sums of NS register slots (slots J0 .. J0+NS-1 of the thread's arrays) over the LDS table into `acc`: complete rows
// are stored by the lane, the piece before the first row end gets the carry of the earlier lanes (segmented wave scan), the
// open tail of the wave goes to wcarry[w].  Rows end only after slots ESTEP-1 (mod ESTEP).
// Round 5 (profiles/r05_den_sq_pmc.txt: 535 VALU + 361 SALU + 104 LDS instructions per wave and frame, two waves per SIMD --
// the passes are bound by instruction issue, not by the LDS: it is busy 20 % of a frame, 44 % of that in bank conflicts):
//  * `addr` holds ABSOLUTE LDS byte addresses (table base + 4 x index, formed once per task): a gather is ds_read_b32 on the
//    register as it stands.  The compiler had hoisted the unpacking of the 16-bit offsets out of the frame loop anyway (64
//    VGPRs) but added the table base to every one of them in every frame;
//  * the row-end mask word is made opaque once per pass.  With a loop-invariant word the compiler precomputed the 32 exec
//    masks of a thread's row-end places, kept them in VGPR lanes (SGPR spills) and paid 2 v_readlane + 2 SALU to fetch
//    each, 6 VALU + 4 SALU per place; now a place is v_add_co (the mask IS vcc) + s_and_saveexec + 3 VALU + restore;
//  * the current row is a byte address that advances by 4, not an index shifted and added to a base per store.
//  * (PK2_DP2_PASS_ASM, default) gathers, row-end places and waits are inline asm.  Left to the compiler, every group of 8
//    multiply-adds of pass A began with s_waitcnt lgkmcnt(0) -- behind the gathers issued ahead for the NEXT group, which
//    were therefore waited for on the spot (the in-pass copy is an asm statement / a conditional block the compiler's
//    wait-count bookkeeping does not see through).  LDS operations complete in order: the wait for group g may leave the
//    row-end stores of group g-1 and the 8 gathers of group g+1 in flight, and says so (lgkmcnt(8 + 8 / ESTEP)).
#ifndef PK2_DP2_PASS_ASM
#define PK2_DP2_PASS_ASM 1
#endif
__device__ __forceinline__ void lds_gather8(const uint32_t* ad, float (&a)[8]) {
  asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %9\n\tds_read_b32 %2, %10\n\tds_read_b32 %3, %11\n\t"
               "ds_read_b32 %4, %12\n\tds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %15"
               : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7])
               : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]) : "memory");
}
// s_waitcnt lgkmcnt(N) that the uses of a[] cannot be scheduled ahead of.
template <int N>
__device__ __forceinline__ void lds_wait8(float (&a)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
               : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait4(float& a0, float& a1, float& a2, float& a3) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "n"(N) : "memory");
}
// A row-end place: the top bit of m says whether this lane's row ends here; if so its sum is stored, the row address moves on.
__device__ __forceinline__ void row_end_place(unsigned& m, uint32_t& cb, float& sum) {
  uint64_t save;
  asm volatile("v_add_co_u32 %0, vcc, %0, %0\n\ts_and_saveexec_b64 %3, vcc\n\tds_write_b32 %1, %2\n\t"
               "v_add_u32 %1, 4, %1\n\tv_mov_b32 %2, 0\n\ts_or_b64 exec, exec, %3"
               : "+v"(m), "+v"(cb), "+v"(sum), "=&s"(save) : : "vcc", "scc", "memory");
}

// PACKED (the orderings with streamed pieces, round 5): `addr` holds the offsets as the layout stores them -- two 16-bit table
// indices to a word, kPK / 2 registers -- and `base` the table's LDS address; an address is formed when its gather is
// issued.  The 32 registers that frees are what the pieces in flight need (the streaming kernels spilled 50-90 registers per
// thread with 64 absolute addresses + three pieces); the two extra VALU instructions per slot are free: the passes are
// not bound by instruction issue.
template <bool PACKED, int J0>
__device__ __forceinline__ void slot_addresses(const uint32_t (&addr)[kPK], uint32_t base, int j0, uint32_t (&out)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int s = J0 + j0 + j;
    if (PACKED) {
      uint32_t pk = addr[s >> 1];
      if ((s & 1) == 0) asm volatile("" : "+v"(pk));      // (opaque per use: the compiler would hoist the unpacking out of the frame loop -- 64 registers again)
      out[j] = base + ((s & 1) ? (pk >> 16) << 2 : (pk & 0xffffu) << 2);
    } else {
      out[j] = addr[s];
    }
  }
}
template <int ESTEP, int J0, int NS, bool PACKED, typename DMA>
__device__ __forceinline__ void pass_rows(const float (&prob)[kPK], const uint32_t (&addr)[kPK], uint32_t base, uint32_t ends, int frow,
                                          float* acc, float* wcarry, const DMA& dma) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float sum = 0.f;
  uint32_t cb = lds_addr(acc) + 4u * (uint32_t)frow;
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < NS / ESTEP; ++k) packed |= ((ends >> (k * ESTEP + ESTEP - 1)) & 1u) << k;
  unsigned m = __builtin_bitreverse32(packed);
  asm volatile("" : "+v"(m));
  float a[2][8];
#if PK2_DP2_PASS_ASM
  constexpr int NG = NS / 8;
  constexpr int kStores = 8 / ESTEP;                 // row-end stores of one group
  uint32_t ad[8];
  slot_addresses<PACKED, J0>(addr, base, 0, ad);
  lds_gather8(ad, a[0]);
  dma(0);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) { slot_addresses<PACKED, J0>(addr, base, 8 * (g + 1), ad); lds_gather8(ad, a[(g + 1) & 1]); dma(2 * g + 2); }
    // younger than the gathers of group g: the stores of group g-1, the gathers of group g+1
    constexpr int kMost = 15;
    if (g == 0) lds_wait8<(NG > 1 ? 8 : 0)>(a[0]);
    else if (g + 1 < NG) lds_wait8<(8 + kStores < kMost ? 8 + kStores : kMost)>(a[g & 1]);
    else lds_wait8<kStores>(a[g & 1]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int jj = 8 * g + j;
      sum = fmaf(a[g & 1][j], prob[J0 + jj], sum);
      if ((jj + 1) % ESTEP == 0) row_end_place(m, cb, sum);
    }
    dma(2 * g + 1);
  }
#else
  auto gather = [&](int j0, float (&a)[8]) {
    uint32_t ad2[8];
    slot_addresses<PACKED, J0>(addr, base, j0, ad2);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = lds_load(ad2[j]);
  };
  gather(0, a[0]);
  dma(0);
#pragma unroll
  for (int g = 0; g < NS / 8; ++g) {
    if (g + 1 < NS / 8) { gather(8 * (g + 1), a[(g + 1) & 1]); dma(2 * g + 2); }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int jj = 8 * g + j;
      sum = fmaf(a[g & 1][j], prob[J0 + jj], sum);
      if ((jj + 1) % ESTEP == 0) {
        if (__builtin_add_overflow(m, m, &m)) { lds_store(cb, sum); cb += 4u; sum = 0.f; }
      }
    }
    dma(2 * g + 1);
  }
#endif
  float x[1] = {sum};
  int fl = ends != 0u ? 1 : 0;
  seg_scan_step<1, 0x111, 0xf>(x, fl);
  seg_scan_step<1, 0x112, 0xf>(x, fl);
  seg_scan_step<1, 0x114, 0xf>(x, fl);
  seg_scan_step<1, 0x118, 0xf>(x, fl);
  seg_scan_step<1, 0x142, 0xa>(x, fl);
  seg_scan_step<1, 0x143, 0xc>(x, fl);
  const float cin = dpp_f<0x138, 0xf>(x[0]);     // wave_shr:1 -- lane 0 receives 0
  if (ends != 0u) acc[frow] += cin;               // the lane's own first row end (stored above by this lane)
  if (lane == 63) wcarry[w] = x[0];
}
template <int J0, bool PACKED, typename DMA>
__device__ __forceinline__ void pass_rows_any(int estep, const float (&prob)[kPK], const uint32_t (&addr)[kPK], uint32_t base,
                                              uint32_t ends, int frow, float* acc, float* wcarry, const DMA& dma) {
  switch (estep) {
    case 8: pass_rows<8, J0, kQ, PACKED>(prob, addr, base, ends, frow, acc, wcarry, dma); break;
    case 4: pass_rows<4, J0, kQ, PACKED>(prob, addr, base, ends, frow, acc, wcarry, dma); break;
    case 2: pass_rows<2, J0, kQ, PACKED>(prob, addr, base, ends, frow, acc, wcarry, dma); break;
    default: pass_rows<1, J0, kQ, PACKED>(prob, addr, base, ends, frow, acc, wcarry, dma); break;
  }
}

// A streamed piece: kSP slots per thread, read from memory in every frame (two 16-byte loads of probabilities, one of packed
// LDS offsets, the row-end bits).  The sums ADD to `acc` (the rows may have slots in other segments; a barrier separates the
// segments); the value a row end adds to is read from LDS one row end ahead, so the add does not wait for it.
// (round 5: the members are native vectors, not arrays -- with float prob[8] / uint32_t idx2[4] the compiler kept every
// Piece in SCRATCH MEMORY: the loads of a piece were stored to the stack, copied from slot to slot there ("cur = nxt") and read
// back one field at a time, ~60 scratch operations per piece and frame)
typedef float __attribute__((ext_vector_type(4))) f32x4;
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4v;
struct Piece { f32x4 p0, p1; u32x4v ix; uint32_t ends; };
#ifndef PK2_ST_SADDR
#define PK2_ST_SADDR 0
#endif
#ifndef PK2_ST_OPAQUE
#define PK2_ST_OPAQUE 1
#endif
#ifndef PK2_ST_PRIME_LATE
#define PK2_ST_PRIME_LATE 0
#endif
__device__ __forceinline__ void piece_load(CDev2& o, int piece, Piece& q) {
  static_assert(kSP == 8, "a piece is two float4 of probabilities and one uint4 of packed offsets per thread");
#if PK2_ST_SADDR
  // uniform base (SGPR pair) + one 32-bit byte offset per thread
  const uint32_t e4 = ((uint32_t)piece * kPT + threadIdx.x) * 4u;           // byte offset of the thread's row-end word
  typedef __attribute__((address_space(1))) const char gchar;
  const __attribute__((address_space(1))) f32x4* sp = (const __attribute__((address_space(1))) f32x4*)((gchar*)o.sprob + e4 * 8u);
  const __attribute__((address_space(1))) u32x4v* si = (const __attribute__((address_space(1))) u32x4v*)((gchar*)o.sidx2 + e4 * 4u);
  const __attribute__((address_space(1))) uint32_t* se = (const __attribute__((address_space(1))) uint32_t*)((gchar*)o.sends + e4);
  q.p0 = sp[0]; q.p1 = sp[1];
  q.ix = si[0];
  q.ends = se[0];
#else
  const size_t at = (size_t)piece * kPT + threadIdx.x;
  const __attribute__((address_space(1))) f32x4* sp = (const __attribute__((address_space(1))) f32x4*)o.sprob + at * (kSP / 4);
  const __attribute__((address_space(1))) u32x4v* si = (const __attribute__((address_space(1))) u32x4v*)o.sidx2 + at;
  const __attribute__((address_space(1))) uint32_t* se = (const __attribute__((address_space(1))) uint32_t*)o.sends;
  q.p0 = sp[0]; q.p1 = sp[1];
  q.ix = si[0];
  q.ends = se[at];
#endif
}
// Round 5: the piece's arithmetic in the form of the resident passes (gathers and waits as inline asm, a row-end place as
// v_add_co + s_and_saveexec + one LDS instruction) -- with ds_add_f32 where a pass stores: the sums ADD to the row, and the
// LDS adds them itself instead of the lane reading the old value one row end ahead, waiting for it (lgkmcnt(0), 2 multiply-
// adds behind the read) and storing the sum.  Same value: old + sum either way.
__device__ __forceinline__ void row_end_add(unsigned& m, uint32_t& cb, float& sum) {
  uint64_t save;
  asm volatile("v_add_co_u32 %0, vcc, %0, %0\n\ts_and_saveexec_b64 %3, vcc\n\tds_add_f32 %1, %2\n\t"
               "v_add_u32 %1, 4, %1\n\tv_mov_b32 %2, 0\n\ts_or_b64 exec, exec, %3"
               : "+v"(m), "+v"(cb), "+v"(sum), "=&s"(save) : : "vcc", "scc", "memory");
}
template <int ESTEP>
__device__ __forceinline__ void piece_rows(const Piece q, uint32_t base, uint32_t& cb, float& sum) {
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < kSP / ESTEP; ++k) packed |= ((q.ends >> (k * ESTEP + ESTEP - 1)) & 1u) << k;
  unsigned m = __builtin_bitreverse32(packed);
  uint32_t ad[8];
#pragma unroll
  for (int j = 0; j < kSP; ++j) {
    const uint32_t pk = q.ix[j >> 1];
    ad[j] = base + ((j & 1) ? (pk >> 16) << 2 : (pk & 0xffffu) << 2);
  }
  float a[8];
  lds_gather8(ad, a);
  // LDS operations complete in order: the first four gathers have landed when four operations are outstanding; the second
  // wait leaves the row-end adds of the first half in flight (one instruction per place, whatever its exec mask)
  lds_wait4<4>(a[0], a[1], a[2], a[3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sum = fmaf(a[j], q.p0[j], sum);
    if ((j + 1) % ESTEP == 0) row_end_add(m, cb, sum);
  }
  lds_wait4<(ESTEP <= 4 ? 4 / ESTEP : 0)>(a[4], a[5], a[6], a[7]);
#pragma unroll
  for (int j = 4; j < 8; ++j) {
    sum = fmaf(a[j], q.p1[j & 3], sum);
    if ((j + 1) % ESTEP == 0) row_end_add(m, cb, sum);
  }
}
__device__ __forceinline__ void piece_rows_any(int estep, const Piece q, uint32_t base, uint32_t& cb, float& sum) {
  switch (estep) {
    case 8: piece_rows<8>(q, base, cb, sum); break;
    case 4: piece_rows<4>(q, base, cb, sum); break;
    case 2: piece_rows<2>(q, base, cb, sum); break;
    default: piece_rows<1>(q, base, cb, sum); break;
  }
}

// The streamed segment of chunk c: pieces [p0, p1) of this rank.  Round 5: the pieces of a rank form ONE sequence per frame
// (segment after segment), and kPieceDepth of them travel in registers: st.r0 is the piece to be summed now, st.r1 is on its
// way.  (With one piece ahead a piece's loads were issued one piece's arithmetic -- ~0.2 us -- before they were needed,
// against ~1 us of L2 latency under load: every piece waited, 1.15 us each in the phase timers of a 1.5 M-arc graph.)  The
// first two pieces of a frame are requested in the frame before, as soon as its last segment has emptied the registers
// (frame_rows: the epilogue, the poll, the copy of chunk 0 and pass A lie between the request and the use); the sequence does not wrap,
// and the barriers between the request and the epilogue's stores order LDS only.  `frow` -- the row a thread's first slot
// of the segment belongs to -- is a constant of the task and comes from a register (it used to be a global load at the head
// of every segment of every frame, with the LDS read of that row's running sum behind it).
constexpr int kPieceDepth = 2;
constexpr int kPieceLoads = 4;       // memory instructions of piece_load (two x4 of probabilities, one x4 of offsets, the row ends)
struct StreamState { Piece r0, r1; int pb[kMaxChunks + 1]; int frow[2]; int pend; };      // (named members, not an array: registers)
static_assert(kPieceDepth == 2, "StreamState holds two pieces");
__device__ __forceinline__ void stream_init(CDev2& o, int rank, StreamState& st) {
#pragma unroll
  for (int c = 0; c <= kMaxChunks; ++c) st.pb[c] = __builtin_amdgcn_readfirstlane(o.pbeg[rank * (kMaxChunks + 1) + c]);
  const __attribute__((address_space(1))) int32_t* sfr = (const __attribute__((address_space(1))) int32_t*)o.sfirst_row;
#pragma unroll
  for (int c = 0; c < 2; ++c) st.frow[c] = st.pb[c + 1] > st.pb[c] ? sfr[((size_t)rank * kMaxChunks + c) * kPT + threadIdx.x] : 0;      // (chunks 2.. : read per frame -- rare, and six registers more cost spills in the passes)
  st.pend = st.pb[0];            // = pb[K], picked with static indices (a dynamic index would keep the whole struct in scratch memory)
#pragma unroll
  for (int c = 1; c <= kMaxChunks; ++c) st.pend = c == o.K ? st.pb[c] : st.pend;
}
__device__ __forceinline__ void stream_prime(CDev2& o, StreamState& st) {
  const int pfirst = st.pb[0], pend = st.pend;
  if (pfirst < pend) piece_load(o, pfirst, st.r0);
  if (pfirst + 1 < pend) piece_load(o, pfirst + 1, st.r1);
}
// Returns the number of pieces it requested (each kPieceLoads memory instructions, the youngest this wave has in flight).
__device__ __forceinline__ int streamed_segment(CDev2& o, int rank, int c, int p0, int p1, StreamState& st, const Lds2& L) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int pend = st.pend;
  const int frow = c < 2 ? st.frow[c < 2 ? c : 0]
                         : ((const __attribute__((address_space(1))) int32_t*)o.sfirst_row)[((size_t)rank * kMaxChunks + c) * kPT + tid];
  float sum = 0.f;
  int issued = 0;
  uint32_t had = 0;
  uint32_t cb = lds_addr(L.accS) + 4u * (uint32_t)frow;
  const uint32_t base = lds_addr(L.table);
  for (int p = p0; p < p1; ++p) {
    had |= st.r0.ends;
    piece_rows_any(o.estep, st.r0, base, cb, sum);
    st.r0 = st.r1;               // (the request goes out BEHIND the piece's arithmetic, into the registers it has freed: a
    if (p + kPieceDepth < pend) { piece_load(o, p + kPieceDepth, st.r1); ++issued; }      // third piece in registers cost spills)
  }
  float x[1] = {sum};
  int fl = had != 0u ? 1 : 0;
  seg_scan_step<1, 0x111, 0xf>(x, fl);
  seg_scan_step<1, 0x112, 0xf>(x, fl);
  seg_scan_step<1, 0x114, 0xf>(x, fl);
  seg_scan_step<1, 0x118, 0xf>(x, fl);
  seg_scan_step<1, 0x142, 0xa>(x, fl);
  seg_scan_step<1, 0x143, 0xc>(x, fl);
  const float cin = dpp_f<0x138, 0xf>(x[0]);
  if (had != 0u) L.accS[frow] += cin;
  if (lane == 63) L.wcarry[(2 + c) * kPW + w] = x[0];
  return issued;
}

// The frame's row sums once its words are valid: both resident passes and every streamed segment over the vector at `src`.
// On return a barrier has NOT been passed yet: the caller's __syncthreads() precedes the first read of the sums.
struct FrameRegs {
  float prob[kPK]; uint32_t addr[kPK]; uint32_t base; uint32_t endsA, endsB; int frowA, frowB;      // addr: absolute LDS byte addresses, or (streaming orderings) addr[0 .. kPK/2) = the packed offsets and base = the table
};
// STREAM = false: the rank-independent fact "this ordering has no streamed piece at all" compiled in -- no piece registers,
// no segment barriers, chunks beyond the two resident ones cannot exist.
struct NoStream { };
struct RowSpan { int nrows, uncA, ncA, uncB, ncB; };
// `stage`: the caller's LDS staging of values it prefetched from memory (x of own rows).  It runs behind the wait for chunk 1,
// where every older memory operation has completed anyway: placed before the copies, the wait for those prefetches -- and
// with it, vmcnt being in order, for the previous frame's history stores -- would delay the copies.
template <bool STREAM, typename ST, typename STAGE>
__device__ __forceinline__ void frame_rows(CDev2& o, cgfloat* src, int rank, const RowSpan& rs, const FrameRegs& r,
                                           const DmaPlan& plan, ST& st, const Lds2& L, DpTimers& dp_, STAGE stage,
                                           bool chunk0_issued = false, bool prime_next = false) {
  const int tid = threadIdx.x;
  if constexpr (STREAM) {
    // the streamed segments add up in accS; compact rows past a truncated resident list get no store from their pass
    for (int q = tid; q < rs.nrows; q += kPT) L.accS[q] = 0.f;
    for (int q = rs.uncA + tid; q < rs.ncA; q += kPT) L.accA[q] = 0.f;
    for (int q = rs.uncB + tid; q < rs.ncB; q += kPT) L.accB[q] = 0.f;
  }
  if (!chunk0_issued) dma_chunk(src, o, 0, L.table, rank);      // (else: copied while the poll was under way)
  DP_T(0);
  wait_vm(0);                            // chunk 0 has landed (this wave's part)
  lds_only_barrier();
  DP_T(1);
  DP_TLF(2);
  Chunk1Dma c1;
  {
    const uint64_t sb = (uint64_t)(src + o.cbeg[1]);
    c1.src = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sb >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sb);
    c1.pl = &plan;
  }
  // (round 4) two chunks that take turns in ONE buffer -- a vector of up to twice the LDS table: chunk 1 may only be
  // copied once every wave has finished pass A (and the streamed segment that gathers from chunk 0)
  const bool shared = o.K == 2 && o.lds_off[1] == o.lds_off[0] && o.cbeg[2] > o.cbeg[1];
  if (shared) {
    pass_rows_any<0, STREAM>(o.estep, r.prob, r.addr, r.base, r.endsA, r.frowA, L.accA, L.wcarry, NoDma());
  } else {
    pass_rows_any<0, STREAM>(o.estep, r.prob, r.addr, r.base, r.endsA, r.frowA, L.accA, L.wcarry, c1);
    for (int j = 8; j * kPW < plan.rows; ++j) c1(j);     // (a chunk 1 of more than 64 rows: the rest, stalling at the issue)
  }
  DP_T(2);
  DP_TLF(3);
  if constexpr (STREAM) {
    const int (&pb)[kMaxChunks + 1] = st.pb;
    const int K = o.K;
    int young = 0;
    if (pb[1] > pb[0]) young = streamed_segment(o, rank, 0, pb[0], pb[1], st, L);
    DP_T(8);
    if (shared) { lds_only_barrier(); dma_chunk(src, o, 1, L.table, rank); young = 0; }
    // chunk 1 has landed (this wave's part) -- the pieces requested behind its copy (the first ones of the next segment) stay
    // in flight through pass B: the barrier orders LDS only (a __syncthreads() would wait for vmcnt(0))
    wait_vm(kPieceLoads * (young < kPieceDepth ? young : kPieceDepth));
    stage();
    lds_only_barrier();                    // chunk 1 is complete, and buffer 0 is free
    DP_T(3);
    if (K > 2) dma_chunk(src, o, 2, L.table, rank);
    pass_rows_any<kQ, STREAM>(o.estep, r.prob, r.addr, r.base, r.endsB, r.frowB, L.accB, L.wcarry + kPW, NoDma());
    DP_T(4);
    // (consecutive segments add to the same rows of accS: the barrier between two chunks separates them)
    if (pb[2] > pb[1]) streamed_segment(o, rank, 1, pb[1], pb[2], st, L);
#pragma unroll
    for (int c = 2; c < kMaxChunks; ++c) {       // (static indices: a dynamic one would put the arrays of `st` into scratch memory)
      if (c >= K) break;
      wait_vm(0);
      __syncthreads();                     // chunk c is complete, and the buffer of chunk c-1 is free
      if (c + 1 < K) dma_chunk(src, o, c + 1, L.table, rank);
      if (pb[c + 1] > pb[c]) streamed_segment(o, rank, c, pb[c], pb[c + 1], st, L);
    }
    // the next frame's first pieces: requested now, into the registers the last segment has emptied -- they flow in while
    // the epilogue works on LDS (the caller's barrier orders LDS only); requested behind the publication instead, the
    // poll's answers queued behind them: +1.5 us per frame between "published" and "words valid" in the phase timers
    if (prime_next && !PK2_ST_PRIME_LATE) stream_prime(o, st);
  } else {
    if (shared) { lds_only_barrier(); dma_chunk(src, o, 1, L.table, rank); }
    wait_vm(0);
    stage();
    __syncthreads();                       // chunk 1 is complete
    DP_T(3);
    pass_rows_any<kQ, STREAM>(o.estep, r.prob, r.addr, r.base, r.endsB, r.frowB, L.accB, L.wcarry + kPW, NoDma());
  }
  if constexpr (STREAM) DP_T(9); else DP_T(4);
}

// Value of rank-local row q after the frame's passes: its entries in the two compact row arrays (and the streamed one) plus
// the wave carry-outs that belong to it.  `cmask` (a constant of the task): the carries whose row is one of this thread's
// rows -- usually none.
template <bool STREAM>
__device__ __forceinline__ float row_val(const Lds2& L, uint64_t cmask, int q) {
  const int a = L.mapA[q], b = L.mapB[q];
  float v = (a >= 0 ? L.accA[a] : 0.f) + (b >= 0 ? L.accB[b] : 0.f);
  if constexpr (STREAM) v += L.accS[q];
  uint64_t m = cmask;
  while (m) {
    const int k = __builtin_ctzll(m);
    m &= m - 1;
    if (L.wcrow[k] == q) v += L.wcarry[k];
  }
  return v;
}

template <int NW>
__device__ __forceinline__ void block_sum2(float& u, float& v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u = wave_sum_dpp(u); v = wave_sum_dpp(v);
  __syncthreads();
  if (lane == 0) { red[w] = u; red[NW + w] = v; }
  __syncthreads();
  float a = 0.f, b = 0.f;
  for (int k = 0; k < NW; ++k) { a += red[k]; b += red[NW + k]; }
  u = a; v = b;
}

template <bool PACKED>
__device__ __forceinline__ void load_frame_regs(CDev2& o, int rank, FrameRegs& r, const float* table) {
  const int tid = threadIdx.x;
  const uint32_t base = lds_addr(table);
  r.base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
  for (int j = 0; j < kPK; ++j) r.prob[j] = o.prob[((size_t)rank * kPK + j) * kPT + tid];
#pragma unroll
  for (int j = 0; j < kPK / 2; ++j) {
    const uint32_t pk = o.idx2[((size_t)rank * (kPK / 2) + j) * kPT + tid];
    if (PACKED) { r.addr[j] = pk; continue; }
    r.addr[2 * j] = base + ((pk & 0xffffu) << 2); r.addr[2 * j + 1] = base + ((pk >> 16) << 2);
    // (opaque: otherwise the compiler keeps the offsets and adds the -- uniform -- base again at every use)
    asm volatile("" : "+v"(r.addr[2 * j]), "+v"(r.addr[2 * j + 1]));
  }
  r.endsA = o.ends[((size_t)rank * 2 + 0) * kPT + tid]; r.endsB = o.ends[((size_t)rank * 2 + 1) * kPT + tid];
  r.frowA = o.first_row[((size_t)rank * 2 + 0) * kPT + tid]; r.frowB = o.first_row[((size_t)rank * 2 + 1) * kPT + tid];
}

// alpha recursion of sequence g (T frames).
template <bool STREAM, int PSPT>
__device__ __noinline__ void run_fwd2(CParams2* pp_, DenPersistCtl* ctl_, DenPersistCtl::Team* team_,
                                      unsigned* nbar, int g_, int T_, int rank_, float* ring_, float* pring_) {
  CParams2* pp = uni(pp_);
  DenPersistCtl* ctl = uni(ctl_); DenPersistCtl::Team* team = uni(team_);
  const int g = uni(g_), T = uni(T_), rank = uni(rank_);
  ring_ = uni(ring_); pring_ = uni(pring_);
  CParams2& p = *pp;
  CDenParams& d = p.d;
  CDev2& o = p.fwd;
  const Lds2 L = carve_lds2(p.tfloats, p.cap, p.rowarrays);
  gfloat* ring = G(ring_); gfloat* pring = G(pring_);
  const int tid = threadIdx.x;
  const int S = d.S, V = d.V, Vo = d.Vo;
  const bool sep = d.alphav != d.alpha;
  FrameRegs fr;
  load_frame_regs<STREAM>(o, rank, fr, L.table);
  DmaPlan plan;
  plan.init(o, L.table, rank);
  const int row0 = o.row_begin[rank], nrows = o.row_begin[rank + 1] - row0;
  const int g0 = o.grp_begin[rank], ngrp = o.grp_begin[rank + 1] - g0;
  RowSpan rs;
  rs.nrows = nrows; rs.uncA = o.uncovered[rank * 2]; rs.uncB = o.uncovered[rank * 2 + 1];
  rs.ncA = o.ncomp[rank * 2]; rs.ncB = o.ncomp[rank * 2 + 1];
  for (int r = tid; r < nrows; r += kPT) { L.mapA[r] = o.rmap[row0 + r]; L.mapB[r] = o.rmap[(size_t)o.num_rows + row0 + r]; }
  typename std::conditional<STREAM, StreamState, NoStream>::type st;
  if constexpr (STREAM) stream_init(o, rank, st);
  for (int r = tid; r < nrows; r += kPT) L.leak[r] = o.row_leak[row0 + r];
  const bool xg = p.xgather != 0;
  if (xg) {
    for (int r = tid; r < nrows; r += kPT) L.pdfv[r] = (short)p.vpdf[row0 + r];
    for (int r = tid; r < ngrp; r += kPT) L.pdfl[r] = (short)p.loop_pdf[g0 + r];
  }
  if (tid < kSegs * kPW) L.wcrow[tid] = o.wcrow[(size_t)rank * kSegs * kPW + tid];
#if PK2_DP2_POLLCOPY
  need_masks(L, o, [&](int r) { return (int)o.grp_begin[r]; });        // rank r publishes the states of its groups
#endif
  int st_lo[PSPT], st_hi[PSPT], st_o[PSPT]; float st_pl[PSPT], st_pi[PSPT]; bool st_ok[PSPT];
#pragma unroll
  for (int i = 0; i < PSPT; ++i) {
    const int r = tid + i * kPT;
    st_ok[i] = r < ngrp;
    st_lo[i] = st_hi[i] = st_o[i] = 0; st_pl[i] = st_pi[i] = 0.f;
    if (st_ok[i]) {
      const int dd = g0 + r;
      st_lo[i] = d.voff[dd] - row0; st_hi[i] = d.voff[dd + 1] - row0; st_o[i] = d.ooff[dd];
      st_pl[i] = d.loop_prob[dd]; st_pi[i] = d.pi[dd];
    }
  }
  // own words of the three frame slots start as "not yet written"
  if (tid < 3 * kPWords) st_agent(pring + ((tid / kPWords) * kPR + rank) * kPWords + tid % kPWords, __uint_as_float(kRingSentinel));
  if (!team_barrier(ctl, team, nbar, L.abort)) return;
  uint64_t cmask = 0;
  for (int k = 0; k < kSegs * kPW; ++k) {
    const int q = L.wcrow[k];
    bool mine = false;
#pragma unroll
    for (int i = 0; i < PSPT; ++i) mine = mine || (st_ok[i] && q >= st_lo[i] && q < st_hi[i]);
    if (q >= 0 && mine) cmask |= 1ull << k;
  }

  const size_t f0 = (size_t)g * (d.Tmax + 1);
  cgfloat* xv_g = G(p.xv) + (size_t)g * d.Tmax * V;
  cgfloat* xl_g = G(d.xl) + (size_t)g * d.Tmax * S;
  // x of the own rows (xr) and of the own states' self-loops (xlr) for the NEXT frame are loaded as soon as the frame has
  // staged its xr (round 4) -- into xr itself and into a second set xln, which becomes xlr at the end of the frame.  They
  // used to be issued at the very end of a frame: vmcnt counts in order, so the polling wave's first poll came back behind
  // its own loads out of HBM -- every rank, even the last to arrive with all 32 words long valid, spent 1.04 us from the top
  // of a frame to "words valid" against 0.2 us of an L2 round trip (timeline of all ranks, -DPK2_DP_PROFILE; a timing-only
  // build whose wave 0 skipped its end-of-frame memory work ran the call 6.8 % faster).
  // (round 4: x comes from the frame's exp(logits) row of P entries, gathered by the pdfs kept in LDS -- 24 KB per frame
  // that the 32 ranks of the team share in L2 -- instead of from copies expanded to V virtual states and S states per frame:
  // 2 x 283 MB written by den_exp_states_lds and read back here per call on the bench graph)
  cgfloat* xp_g = G(p.xv) + (size_t)g * d.Tmax * d.P;
  float xr[PSPT], xlr[PSPT], xln[PSPT];
  auto prefetch = [&](int t) {
    if (xg) {
      cgfloat* row = xp_g + (size_t)t * d.P;
#pragma unroll
      for (int i = 0; i < PSPT; ++i) {
        const int r = per_frame<STREAM && PK2_ST_OPAQUE>(tid + i * kPT);
        const int pv = r < nrows ? L.pdfv[r] : -1, pl = st_ok[i] ? L.pdfl[r] : -1;
        xr[i] = r < nrows ? (pv >= 0 ? row[pv] : 1.f) : 0.f;
        xln[i] = st_ok[i] ? (pl >= 0 ? row[pl] : 1.f) : 0.f;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < PSPT; ++i) {
      const int r = per_frame<STREAM && PK2_ST_OPAQUE>(tid + i * kPT);
      if constexpr (STREAM && PK2_ST_OPAQUE) {
        xr[i] = r < nrows ? ld_uniform_base(xv_g + ((size_t)t * V + row0), r) : 0.f;
        xln[i] = st_ok[i] ? ld_uniform_base(xl_g + ((size_t)t * S + g0), r) : 0.f;
      } else {
        xr[i] = r < nrows ? once_load<!STREAM>(&xv_g[(size_t)t * V + row0 + r]) : 0.f;
        xln[i] = st_ok[i] ? once_load<!STREAM>(&xl_g[(size_t)t * S + g0 + r]) : 0.f;
      }
    }
  };
  prefetch(0);
#pragma unroll
  for (int i = 0; i < PSPT; ++i) xlr[i] = xln[i];
  if constexpr (STREAM) stream_prime(o, st);
  float own_a[PSPT];
#pragma unroll
  for (int i = 0; i < PSPT; ++i) own_a[i] = st_pi[i];          // alpha[0, .] = pi
  // the history the parallel passes read, frame tt: alpha[tt + 1] (own_a, until the next epilogue), the rows in L.aux
  float hist_loop[PSPT], hist_loc = 0.f;
#pragma unroll
  for (int i = 0; i < PSPT; ++i) hist_loop[i] = 0.f;
  auto history = [&](int tt) {
    gfloat* alpha_n = G(d.alpha) + (f0 + tt + 1) * (size_t)S;
    gfloat* alphav_n = G(d.alphav) + (f0 + tt + 1) * (size_t)Vo;
    if (tid == 0) G(d.apart)[(f0 + tt + 1) * kPR + rank] = hist_loc;
#pragma unroll
    for (int i = 0; i < PSPT; ++i) {
      if (!st_ok[i]) continue;
      // (the occupancy pass reads alpha per OCCUPANCY state only; the per-state copy has no reader behind this kernel
      // -- 225 MB of stores per call on the bench graph until round 4 -- unless the two arrays are one: Vo == S)
      if (!sep) once_store<!STREAM>(&alpha_n[g0 + tid + i * kPT], own_a[i]);
      if (sep) {
        for (int q = st_lo[i]; q < st_hi[i]; ++q) once_store<!STREAM>(&alphav_n[st_o[i] + q - st_lo[i]], L.aux[q]);
        if (st_pl[i] > 0.f) once_store<!STREAM>(&alphav_n[st_o[i] + st_hi[i] - st_lo[i]], hist_loop[i]);
      }
    }
  };
  Spin spin(ctl);
  DP_T0();
  for (int t = 0; t < T; ++t) {
    float as;
    cgfloat* src;
    DP_TL(0, 0);
    DP_TLSET(0);
    bool chunk0_issued = false;
    if (t == 0) {
      as = d.pi_sum;
      src = G(d.pi);
    } else {
      src = ring + (size_t)(t & 1) * p.rpad;
#if PK2_DP2_POLLCOPY
      chunk0_issued = poll_and_copy0(pring + (size_t)(t % 3) * kPR * kPWords, 1, spin, L, src, o);
#else
      poll_words(pring + (size_t)(t % 3) * kPR * kPWords, 0, 1, spin, L);
#endif
      __syncthreads();
      DP_TL(0, 1);
      if (*L.abort) return;
      as = L.tot[0];
      if (tid == 0) L.abort[8] = 0;        // (every wave has left the poll: the next one starts from "nobody there")
    }
    // the word this rank will publish two frames from now must read "not yet written" by then: reset here, before the
    // copies (no store sits between them and their waits), long before the stores it has to precede
    const bool publish = t + 1 < T;
    if (tid == 0 && publish) st_agent(word_of(pring, t + 2, rank, 0), __uint_as_float(kRingSentinel));
    frame_rows<STREAM>(o, src, rank, rs, fr, plan, st, L, dp_, [&]() {
#pragma unroll
      for (int i = 0; i < PSPT; ++i) {
        const int r = tid + i * kPT;
        if (r < nrows) L.xown[r] = xr[i];
      }
      if (kLateHistory<STREAM> && t > 0) history(t - 1);
      if (publish) prefetch(t + 1);
    }, chunk0_issued, publish);
    DP_TL(0, 4);
    if constexpr (STREAM) lds_only_barrier(); else __syncthreads();      // (the row sums are in LDS; a streaming frame has piece loads in flight)
    if (rank == 0 && tid == 0) G(d.asum)[f0 + t] = as;
    const float lk = d.leaky * as, inv_as = 1.0f / as;
    // rows are virtual states (dst, pdf); a thread per real state sums its rows and adds the peeled self-loop.  First
    // only what the other workgroups wait for: the ring entries, then (once they are in L2) the partial sum.
    gfloat* ring_n = ring + (size_t)((t + 1) & 1) * p.rpad;
    float outv[PSPT], loopv[PSPT], loc = 0.f, unused = 0.f;
#pragma unroll
    for (int i = 0; i < PSPT; ++i) {
      outv[i] = 0.f; loopv[i] = 0.f;
      if (!st_ok[i]) continue;
      float sum = 0.f;
      for (int q = st_lo[i]; q < st_hi[i]; ++q) {
        const float v = (row_val<STREAM>(L, cmask, q) + lk * L.leak[q]) * L.xown[q] * inv_as;
        L.aux[q] = v;          // (the row belongs to this thread alone: kept for the history stores below)
        sum += v;
      }
      if (st_pl[i] > 0.f) { loopv[i] = (own_a[i] + lk * st_pi[i]) * st_pl[i] * xlr[i] * inv_as; sum += loopv[i]; }
      if (publish) ring_store(ring_n + g0 + tid + i * kPT, sum);
      outv[i] = sum;
      own_a[i] = sum;          // alpha[t+1] of the own state: the next frame's loop term
      loc += sum;
    }
    DP_T(5);
    DP_TL(0, 5);
    wait_stores();        // this rank's slice of frame t+1 (and the reset above) is in L2 before its partial sum says so
    block_sum2<kPW>(loc, unused, L.red);
    if (tid == 0 && publish) st_agent(word_of(pring, t + 1, rank, 0), loc);
    if constexpr (STREAM && PK2_ST_PRIME_LATE) { if (publish) stream_prime(o, st); }
    DP_T(6);
    DP_TL(0, 6);
    hist_loc = loc;
#pragma unroll
    for (int i = 0; i < PSPT; ++i) hist_loop[i] = loopv[i];
    if (!kLateHistory<STREAM>) history(t);
#pragma unroll
    for (int i = 0; i < PSPT; ++i) xlr[i] = xln[i];
    DP_T(7);
  }
  if (kLateHistory<STREAM> && T > 0) history(T - 1);
  DP_FLUSH(0);
}

// btilde' recursion of sequence g, T-1 down to 0 (see chain_den_persist.hip: run_bwd for the algebra).
template <bool STREAM, int PSPT>
__device__ __noinline__ void run_bwd2(CParams2* pp_, DenPersistCtl* ctl_, DenPersistCtl::Team* team_,
                                      unsigned* nbar, int g_, int T_, int rank_, float* ring_, float* pring_) {
  CParams2* pp = uni(pp_);
  DenPersistCtl* ctl = uni(ctl_); DenPersistCtl::Team* team = uni(team_);
  const int g = uni(g_), T = uni(T_), rank = uni(rank_);
  ring_ = uni(ring_); pring_ = uni(pring_);
  CParams2& p = *pp;
  CDenParams& d = p.d;
  CDev2& o = p.bwd;
  const Lds2 L = carve_lds2(p.tfloats, p.cap, p.rowarrays);
  gfloat* ring = G(ring_); gfloat* pring = G(pring_);
  const int tid = threadIdx.x;
  const int S = d.S, V = d.V;
  FrameRegs fr;
  load_frame_regs<STREAM>(o, rank, fr, L.table);
  DmaPlan plan;
  plan.init(o, L.table, rank);
  const int row0 = o.row_begin[rank], nrows = o.row_begin[rank + 1] - row0;
  const int vfirst = d.voff[row0], nvirt = d.voff[row0 + nrows] - vfirst;     // own virtual states: a contiguous range
  RowSpan rs;
  rs.nrows = nrows; rs.uncA = o.uncovered[rank * 2]; rs.uncB = o.uncovered[rank * 2 + 1];
  rs.ncA = o.ncomp[rank * 2]; rs.ncB = o.ncomp[rank * 2 + 1];
  for (int r = tid; r < nrows; r += kPT) { L.mapA[r] = o.rmap[row0 + r]; L.mapB[r] = o.rmap[(size_t)o.num_rows + row0 + r]; }
  typename std::conditional<STREAM, StreamState, NoStream>::type st;
  if constexpr (STREAM) stream_init(o, rank, st);
  if (tid < kSegs * kPW) L.wcrow[tid] = o.wcrow[(size_t)rank * kSegs * kPW + tid];
#if PK2_DP2_POLLCOPY
  need_masks(L, o, [&](int r) { return (int)d.voff[o.row_begin[r]]; });      // rank r publishes the virtual states of its states
#endif
  int st_v0[PSPT], st_v1[PSPT]; float st_pl[PSPT], st_pi[PSPT]; bool st_ok[PSPT];
#pragma unroll
  for (int i = 0; i < PSPT; ++i) {
    const int r = tid + i * kPT;
    st_ok[i] = r < nrows;
    st_v0[i] = st_v1[i] = 0; st_pl[i] = st_pi[i] = 0.f;
    if (st_ok[i]) {
      const int s = row0 + r;
      st_v0[i] = d.voff[s] - vfirst; st_v1[i] = d.voff[s + 1] - vfirst; st_pl[i] = d.loop_prob[s]; st_pi[i] = d.pi[s];
    }
  }
  // what an own virtual state contributes to the next frame's sums (virtual states are the forward layout's rows)
  for (int r = tid; r < nvirt; r += kPT) { L.leak[r] = p.fwd.row_leak[vfirst + r]; L.aux[r] = p.fwd.row_psum[vfirst + r]; }
  const bool xg = p.xgather != 0;
  if (xg) {
    for (int r = tid; r < nvirt; r += kPT) L.pdfv[r] = (short)p.vpdf[vfirst + r];
    for (int r = tid; r < nrows; r += kPT) L.pdfl[r] = (short)p.loop_pdf[row0 + r];
  }
  if (tid < 3 * kPWords) st_agent(pring + ((tid / kPWords) * kPR + rank) * kPWords + tid % kPWords, __uint_as_float(kRingSentinel));
  if (!team_barrier(ctl, team, nbar, L.abort)) return;
  uint64_t cmask = 0;
  for (int k = 0; k < kSegs * kPW; ++k) {
    const int q = L.wcrow[k];
    bool mine = false;
#pragma unroll
    for (int i = 0; i < PSPT; ++i) mine = mine || (st_ok[i] && q == tid + i * kPT);
    if (q >= 0 && mine) cmask |= 1ull << k;
  }

  const size_t f0 = (size_t)g * (d.Tmax + 1);
  cgfloat* xv_g = G(p.xv) + (size_t)g * d.Tmax * V;
  cgfloat* xl_g = G(d.xl) + (size_t)g * d.Tmax * S;
  float xl_cur[PSPT], xl_prev[PSPT], xl_next[PSPT], xw[PSPT], bh[PSPT];
  const float cst_last = 1.0f / d.pi_sum + d.leaky;
#pragma unroll
  for (int i = 0; i < PSPT; ++i) { bh[i] = cst_last; xl_cur[i] = 0.f; }
  cgfloat* xp_g = G(p.xv) + (size_t)g * d.Tmax * d.P;       // (x gathered from the frame's exp row: run_fwd2)
  auto prefetch = [&](int t) {
    if (xg) {
      cgfloat* row = xp_g + (size_t)t * d.P;
#pragma unroll
      for (int i = 0; i < PSPT; ++i) {
        const int r = per_frame<STREAM && PK2_ST_OPAQUE>(tid + i * kPT);
        const int pl = st_ok[i] ? L.pdfl[r] : -1, pv = r < nvirt ? L.pdfv[r] : -1;
        xl_next[i] = st_ok[i] ? (pl >= 0 ? row[pl] : 1.f) : 0.f;
        xw[i] = r < nvirt ? (pv >= 0 ? row[pv] : 1.f) : 0.f;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < PSPT; ++i) {
      const int r = per_frame<STREAM && PK2_ST_OPAQUE>(tid + i * kPT);
      if constexpr (STREAM && PK2_ST_OPAQUE) {
        xl_next[i] = st_ok[i] ? ld_uniform_base(xl_g + ((size_t)t * S + row0), r) : 0.f;
        xw[i] = r < nvirt ? ld_uniform_base(xv_g + ((size_t)t * V + vfirst), r) : 0.f;
      } else {
        xl_next[i] = st_ok[i] ? once_load<!STREAM>(&xl_g[(size_t)t * S + row0 + r]) : 0.f;
        xw[i] = r < nvirt ? once_load<!STREAM>(&xv_g[(size_t)t * V + vfirst + r]) : 0.f;
      }
    }
  };
  auto stage_x = [&]() {
#pragma unroll
    for (int i = 0; i < PSPT; ++i) {
      const int r = tid + i * kPT;
      if (r < nvirt) L.xown[r] = xw[i];
    }
  };
  auto emit = [&](int t) {
    gfloat* ring_n = ring + (size_t)(t & 1) * p.rpad + vfirst;
    float pB = 0.f, pU = 0.f;
#pragma unroll
    for (int i = 0; i < PSPT; ++i) {
      if (!st_ok[i]) continue;
      for (int q = st_v0[i]; q < st_v1[i]; ++q) {
        const float w = L.xown[q] * bh[i];
        ring_store(ring_n + q, w);
        pB = fmaf(w, L.leak[q], pB); pU = fmaf(w, L.aux[q], pU);
      }
      if (st_pl[i] > 0.f) { const float lp = st_pl[i] * xl_prev[i] * bh[i]; pB = fmaf(st_pi[i], lp, pB); pU += lp; }
    }
    wait_stores();        // the slice (and the reset of the words two frames on) is in L2 before the sums say so
    block_sum2<kPW>(pB, pU, L.red);
    if (tid < 2) st_agent(word_of(pring, t, rank, tid), tid == 0 ? pB : pU);
    if (tid == 0) {       // the parallel passes recompute c[t-1] from the same shares
      G(d.bpart)[((f0 + t - 1) * kPR + rank) * 2] = pB;
      G(d.bpart)[((f0 + t - 1) * kPR + rank) * 2 + 1] = pU;
    }
  };
  // w[T] = x[T-1, .] * (1 / sum(pi) + leaky)
  // (the loads of frame t - 2 are issued from inside frame t - 1, behind its staging, not at the end of frame t: run_fwd2)
  prefetch(T - 1);
#pragma unroll
  for (int i = 0; i < PSPT; ++i) xl_prev[i] = xl_next[i];
  stage_x();
  __syncthreads();
  emit(T);
#pragma unroll
  for (int i = 0; i < PSPT; ++i) xl_cur[i] = xl_prev[i];
  if (T >= 2) prefetch(T - 2);
#pragma unroll
  for (int i = 0; i < PSPT; ++i) xl_prev[i] = xl_next[i];
  // btilde'[tt, .] of the own states for the parallel passes (run_fwd2: from the next frame's staging point)
  float vs_hist[PSPT];
#pragma unroll
  for (int i = 0; i < PSPT; ++i) vs_hist[i] = 0.f;
  auto history = [&](int tt) {
    gfloat* bx_t = G(d.beta) + (f0 + tt) * (size_t)V * d.brec;
#pragma unroll
    for (int i = 0; i < PSPT; ++i)
      if (st_ok[i]) once_store<!STREAM>(&bx_t[(size_t)(vfirst + st_v0[i]) * d.brec], vs_hist[i]);
  };
  if constexpr (STREAM) stream_prime(o, st);
  Spin spin(ctl);
  DP_T0();
  for (int t = T - 1; t >= 0; --t) {
    DP_TL(1, 0);
    DP_TLSET(1);
    cgfloat* src_t = ring + (size_t)((t + 1) & 1) * p.rpad;
#if PK2_DP2_POLLCOPY
    const bool chunk0_issued = poll_and_copy0(pring + (size_t)((t + 1) % 3) * kPR * kPWords, 2, spin, L, src_t, o);
#else
    const bool chunk0_issued = false;
    poll_words(pring + (size_t)((t + 1) % 3) * kPR * kPWords, 0, 2, spin, L);
#endif
    __syncthreads();
    DP_TL(1, 1);
    if (*L.abort) return;
    const float lB = L.tot[0], lU = L.tot[1];        // sums of btilde'[t, .], known before it is computed
    if (tid == 0) L.abort[8] = 0;
    const bool publish = t > 0;
    // the words this rank will publish two frames from now must read "not yet written" by then: reset here, before the
    // copies, long before the stores they have to precede (the waits of emit cover it)
    if (tid < 2 && publish) st_agent(word_of(pring, t + 2, rank, tid), __uint_as_float(kRingSentinel));   // (t-1) % 3 == (t+2) % 3
    frame_rows<STREAM>(o, src_t, rank, rs, fr, plan, st, L, dp_, [&]() {
      if (publish) stage_x();
      if (kLateHistory<STREAM> && t + 1 < T) history(t + 1);
      if (t >= 2) prefetch(t - 2);          // (xw is free again; xl_next becomes xl_prev at the end of the frame)
    }, chunk0_issued, publish);
    DP_TL(1, 4);
    if constexpr (STREAM) lds_only_barrier(); else __syncthreads();
    // rows are source states: btilde'[t, s] = row + peeled loop, then beta-hat[t, s] with the sums received above
    const float cu = lB + d.wu * lU;
    const float inv_c = cu > 0.f ? 1.0f / cu : 0.f;
    const float lkr = d.leaky * lB * inv_c;
    float vs[PSPT];
#pragma unroll
    for (int i = 0; i < PSPT; ++i) {
      vs[i] = 0.f;
      if (!st_ok[i]) continue;
      float v = row_val<STREAM>(L, cmask, tid + i * kPT);
      if (st_pl[i] > 0.f) v += st_pl[i] * xl_cur[i] * bh[i];
      vs[i] = v;
      bh[i] = v * inv_c + lkr;
    }
    DP_T(5);
    DP_TL(1, 5);
    if (publish) emit(t);
    if constexpr (STREAM && PK2_ST_PRIME_LATE) { if (publish) stream_prime(o, st); }
    DP_T(6);
    DP_TL(1, 6);
#pragma unroll
    for (int i = 0; i < PSPT; ++i) {
      vs_hist[i] = vs[i];
      xl_cur[i] = xl_prev[i];
      xl_prev[i] = xl_next[i];
    }
    if (!kLateHistory<STREAM>) history(t);
    DP_T(7);
  }
  if (kLateHistory<STREAM> && T > 0) history(0);
  DP_FLUSH(1);
}

__global__ void __launch_bounds__(kPT) den_persist2_kernel(const DenPersist2Params* __restrict__ pp_, DenPersistCtl* ctl) {
  CParams2* pp = (CParams2*)pp_;
  CParams2& p = *pp;
  const Lds2 L = carve_lds2(p.tfloats, p.cap, p.rowarrays);
  int& s_abort = L.abort[0];
  int& s_rank = L.abort[1]; int& s_team = L.abort[2]; int& s_xcd = L.abort[3]; int& s_task = L.abort[4];
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_abort = 0;
    const unsigned xcd = den_xcc_id();
    const unsigned slot = __hip_atomic_fetch_add(&ctl->arrive[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_rank = (int)(slot % kPR); s_team = (int)(slot / kPR); s_xcd = (int)xcd;
  }
  __syncthreads();
  if (s_team >= kMaxTeams) return;
  const int rank = s_rank;
  DenPersistCtl::Team* team = &ctl->team[s_xcd][s_team];
  const size_t ti = (size_t)s_xcd * kMaxTeams + s_team;
  float* ring = p.ring + ti * 2 * p.rpad;
  float* pring = p.pring + ti * 3 * kPR * kPWords;
  unsigned nbar = 0;
  for (int iter = 0; iter <= kMaxTasks; ++iter) {
    if (tid == 0) {
      unsigned k;
      if (rank == 0) {
        k = __hip_atomic_fetch_add(&ctl->next_task, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&team->task[iter], k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        Spin spin(ctl);
        unsigned k1;
        while ((k1 = ld_agent_u(G(&team->task[iter]))) == 0u) {
          if (spin.expired()) { s_abort = 1; k1 = 1u << 30; break; }
        }
        k = k1 - 1u;
      }
      s_task = (int)k;
    }
    __syncthreads();
    const int k = s_task;
    if (s_abort || k >= p.ntasks) return;
    const int g = p.task_seq[k], T = p.d.lengths[g];
    if (!team_barrier(ctl, team, &nbar, &s_abort)) return;      // everybody has left the previous recursion
    if (p.task_dir[k] == 2) {
      // A numerator forward-backward (chain_num.h: two waves, everything staged in LDS) in the idle time of a team whose
      // recursions are done -- the recursions are queued longest first, so the teams of the short sequences get here while
      // the longest is still running.  Rank 0's workgroup does it (the LDS of the finished recursion is free); its other
      // waves execute the routine's barriers with it.
      if (rank == 0) {
        if (tid < 128) {
          NumParams np = pp_->np;      // (through the generic pointer: the struct is passed on by reference)
          num_fwd_bwd_two_waves(np, g, den_persist2_smem);
        } else {
#pragma unroll
          for (int q = 0; q < kNumTwoWaveBarriers; ++q) __syncthreads();
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      continue;
    }
    // (PSPT: rows / states / own virtual states a thread handles in the row epilogues -- 2 when no rank has more than
    // 2 * kPT of any of them: the per-state constants then take half the registers)
    if (p.task_dir[k] == 0) {
      // (three entries per thread since round 4: graphs of 33 - 49 k states have 1025 .. 1536 rows per rank)
      if (p.fwd_stream) { if (p.pspt == 2) run_fwd2<true, 2>(pp, ctl, team, &nbar, g, T, rank, ring, pring); else if (p.pspt == 3) run_fwd2<true, 3>(pp, ctl, team, &nbar, g, T, rank, ring, pring); else run_fwd2<true, kPSPT>(pp, ctl, team, &nbar, g, T, rank, ring, pring); }
      else { if (p.pspt == 2) run_fwd2<false, 2>(pp, ctl, team, &nbar, g, T, rank, ring, pring); else if (p.pspt == 3) run_fwd2<false, 3>(pp, ctl, team, &nbar, g, T, rank, ring, pring); else run_fwd2<false, kPSPT>(pp, ctl, team, &nbar, g, T, rank, ring, pring); }
    } else {
      if (p.bwd_stream) { if (p.pspt == 2) run_bwd2<true, 2>(pp, ctl, team, &nbar, g, T, rank, ring, pring); else if (p.pspt == 3) run_bwd2<true, 3>(pp, ctl, team, &nbar, g, T, rank, ring, pring); else run_bwd2<true, kPSPT>(pp, ctl, team, &nbar, g, T, rank, ring, pring); }
      else { if (p.pspt == 2) run_bwd2<false, 2>(pp, ctl, team, &nbar, g, T, rank, ring, pring); else if (p.pspt == 3) run_bwd2<false, 3>(pp, ctl, team, &nbar, g, T, rank, ring, pring); else run_bwd2<false, kPSPT>(pp, ctl, team, &nbar, g, T, rank, ring, pring); }
    }
    __syncthreads();
    if (s_abort) return;
    if (tid == 0 && rank == 0) __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// (round 5: it also leaves the control block zeroed for the next launch on this stream -- one fill kernel less per call)
__global__ void den_persist2_check(DenPersistCtl* ctl, int ntasks, float* den_lp, int n, unsigned* guard_dev, unsigned* guard_host) {
  const bool bad = ctl->abort != 0u || ctl->done != (unsigned)ntasks;
  __syncthreads();
  for (unsigned i = threadIdx.x; i < sizeof(DenPersistCtl) / sizeof(unsigned); i += blockDim.x) reinterpret_cast<unsigned*>(ctl)[i] = 0u;
  if (bad) {
    if (threadIdx.x == 0) persist_guard_raise(guard_dev, guard_host);
    for (int i = threadIdx.x; i < n; i += blockDim.x) den_lp[i] = __uint_as_float(0x7fc00000u);
  }
}

// ----------------------------------------------------------------------------------------
// host
// ----------------------------------------------------------------------------------------
static PerDevice<int> g_den_persist2_state_pd(-1);     // -1: not verified yet, 0: unusable on this device, 1: verified
struct DenPersist2Scratch { DenPersist2Params* params = nullptr; DenPersistCtl* ctl = nullptr; float* ring = nullptr; float* pring = nullptr; int rpad = 0; int ntasks = 0;
                            bool ctl_clean = false;         // the last launch's check kernel has zeroed the control block
                            std::vector<unsigned char> params_host; };      // what `params` holds (a call with the same block skips the store)
static std::map<DevStream, DenPersist2Scratch> g_den2_scratch;

static int den2_rpad(const pk2_den_graph* g) { return (std::max(g->S, g->V) + 255) / 256 * 256 + 256; }
static int den2_tfloats(const pk2_den_graph* g) { return std::max(g->h_p2fwd.tfloats, g->h_p2bwd.tfloats); }

bool den_persist2_fits(const pk2_den_graph* g) {
  return g->h_p2fwd.ok && g->h_p2bwd.ok && g->p2_cap > 0 && den_persist2_lds_bytes(den2_tfloats(g), g->p2_cap, g->p2_rowarrays) <= kDenPersistMaxLds;
}

static bool den_is_8x32() {
  static PerDevice<int> cus_pd(-1); int& cus = cus_pd.ref();
  if (cus < 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus = n;
  }
  return cus == 8 * kPR;
}

// Which form runs: the forced one if it fits; else the second form whenever the whole graph is resident in it (it takes
// every graph the first form takes and is a little faster); when it has to stream pieces or rotate table chunks, whichever of
// it and the launch-per-frame kernels the measured cost model (profiles/r03_den_sweep.txt) predicts to be faster.
int den_persist_version(const pk2_den_graph* g, int N) {
  const char* env = getenv("PK2_DEN_PERSIST");
  const int want = env ? atoi(env) : -1;
  if (want == 0 || !den_use_sx(g) || N < 1 || 2 * N > kMaxTasks || !den_is_8x32()) return 0;
  const bool ok1 = den_persist_wanted(g, N);
  const bool ok2 = g_den_persist2_state_pd.ref() != 0 && den_persist2_fits(g);
  if (want == 1) return ok1 ? 1 : 0;
  if (want == 2) return ok2 ? 2 : 0;
  if (!ok2) return ok1 ? 1 : 0;
  const HostPersist2& f = g->h_p2fwd; const HostPersist2& b = g->h_p2bwd;
  const int pieces = std::max(f.max_pieces, b.max_pieces), K = std::max(f.K, b.K);
  if (pieces == 0 && K == 2) return 2;
  // microseconds per frame, fitted to the sweep: the streaming frame grows with the vector (table copy, row epilogues,
  // the row-indexed sums of the segments) and by ~0.9 per streamed piece of 4096 slots; the frame kernels stream both arc
  // lists once per frame for up to 4 sequences at a time
  const double S = (double)g->S * 1e-3, A = (double)g->A * 1e-6;
  const double us2 = 5.2 + 0.30 * S + 0.9 * pieces;
  const double usf = 6.0 + 0.12 * S + 10.5 * A;
  // 2 N recursions on 8 XCDs (each sequence's two recursions side by side) against ceil(N / 4) groups of frame launches
  const double t2 = us2 * std::max(1.0, 2.0 * N / 8.0), tf = usf * ((N + 3) / 4);
  return t2 <= tf ? 2 : (ok1 ? 1 : 0);
}

int den_persist2_launch(pk2_den_graph* g, const DenParams& dp, const float* xv, const int32_t* lengths_host, int N,
                        hipStream_t stream, bool* ran, const NumDeferred* tail, bool* num_ran, bool xgather) {
  *ran = false;
  if (num_ran) *num_ran = false;
  DenPersist2Scratch& sc = g_den2_scratch[dev_stream(stream)];
  const int rpad = den2_rpad(g);
  if (!sc.ctl) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.ctl), sizeof(DenPersistCtl)));
  if (!sc.params) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.params), sizeof(DenPersist2Params)));
  if (!sc.pring) PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.pring), (size_t)8 * kMaxTeams * 3 * kPR * kPWords * sizeof(float)));
  if (sc.rpad < rpad) {
    if (sc.ring) PK2_HIP(hipFree(sc.ring));
    sc.ring = nullptr; sc.rpad = 0;
    PK2_HIP(hipMalloc(reinterpret_cast<void**>(&sc.ring), (size_t)8 * kMaxTeams * 2 * rpad * sizeof(float)));
    // (entries past a vector's end are copied into LDS with the last 16-byte granule but never gathered: keep them finite)
    PK2_HIP(hipMemsetAsync(sc.ring, 0, (size_t)8 * kMaxTeams * 2 * rpad * sizeof(float), stream));
    sc.rpad = rpad;
  }
  DenPersist2Params p;
  memset(static_cast<void*>(&p), 0, sizeof(p));      // (padding too: the block is compared with the last one stored)
  p.d = dp;
  p.fwd = g->p2fwd; p.bwd = g->p2bwd;
  p.xv = xv; p.ring = sc.ring; p.pring = sc.pring;
  p.vpdf = g->d_vpdf; p.loop_pdf = g->d_loop_pdf; p.xgather = xgather ? 1 : 0; p.rowarrays = g->p2_rowarrays;
  p.rpad = rpad; p.tfloats = den2_tfloats(g); p.cap = g->p2_cap;
  p.pspt = g->p2_cap <= 2 * kPT ? 2 : (g->p2_cap <= 3 * kPT ? 3 : kPSPT);
  p.fwd_stream = (!g->h_p2fwd.sends.empty() || g->h_p2fwd.K > 2) ? 1 : 0;
  p.bwd_stream = (!g->h_p2bwd.sends.empty() || g->h_p2bwd.K > 2) ? 1 : 0;
  std::vector<std::pair<int, int>> order;    // (-T, task id = 2 n + dir), longest first
  for (int n = 0; n < N; ++n)
    if (lengths_host[n] > 0) { order.push_back({-lengths_host[n], 2 * n}); order.push_back({-lengths_host[n], 2 * n + 1}); }
  std::sort(order.begin(), order.end());
  p.ntasks = (int)order.size();
  for (int k = 0; k < kMaxTasks; ++k) { p.task_seq[k] = 0; p.task_dir[k] = 0; }
  for (int k = 0; k < p.ntasks; ++k) { p.task_seq[k] = (short)(order[k].second >> 1); p.task_dir[k] = (unsigned char)(order[k].second & 1); }
  sc.ntasks = 0;
  if (p.ntasks == 0) { *ran = true; return PK2_OK; }
  // The numerator forward-backward of the minibatch (two waves per sequence, LDS-staged) as further tasks behind the
  // recursions: the teams of the short sequences run them while the longest recursion is still going.  Not during the
  // first, verified launch of a process (a fallback would have to undo the posteriors already added to the gradient).
  const bool with_num = tail && tail->valid && tail->stage && num_ran && g_den_persist2_state_pd.ref() == 1 && tail->N == N &&
                        tail->lds <= (size_t)den2_tfloats(g) * sizeof(float) && p.ntasks + N <= kMaxTasks &&
                        !(getenv("PK2_DEN_NUM_RIDE") && atoi(getenv("PK2_DEN_NUM_RIDE")) == 0);
  if (with_num) {
    p.np = tail->p;
    for (int n = 0; n < N; ++n) { p.task_seq[p.ntasks] = (short)n; p.task_dir[p.ntasks] = 2; ++p.ntasks; }
  }
  const size_t lds = den_persist2_lds_bytes(p.tfloats, p.cap, p.rowarrays);
  static PerDevice<bool> attr_pd(false); bool& attr = attr_pd.ref();
  if (!attr) {
    PK2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&den_persist2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr = true;
  }
  if (!sc.ctl_clean) PK2_HIP(hipMemsetAsync(sc.ctl, 0, sizeof(DenPersistCtl), stream));
  sc.ctl_clean = false;
  {
    const unsigned char* pb = reinterpret_cast<const unsigned char*>(&p);
    if (sc.params_host.size() != sizeof(p) || memcmp(sc.params_host.data(), pb, sizeof(p)) != 0) {
      hipLaunchKernelGGL(param_block_store<DenPersist2Params>, dim3(1), dim3(1), 0, stream, p, sc.params);
      sc.params_host.assign(pb, pb + sizeof(p));
    }
  }
  hipLaunchKernelGGL(den_persist2_kernel, dim3(8 * kPR), dim3(kPT), lds, stream, sc.params, sc.ctl);
#ifdef PK2_DP_PROFILE
  { int tot = 0; for (int n = 0; n < N; ++n) tot += lengths_host[n];
    hipLaunchKernelGGL(dp_prof_print, dim3(1), dim3(1), 0, stream, std::max(1, tot / 4), 2); }
#endif
  PK2_LAUNCH_CHECK();
  if (g_den_persist2_state_pd.ref() < 0) {     // first use on this device: every recursion done, nobody timed out?
    DenPersistCtl* h = new DenPersistCtl;
    hipError_t e = hipMemcpyAsync(h, sc.ctl, sizeof(DenPersistCtl), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    const bool ok = e == hipSuccess && h->abort == 0u && h->done == (unsigned)p.ntasks;
    delete h;
    if (e != hipSuccess) { set_error("den_persist2: %s", hipGetErrorString(e)); return PK2_ERR_HIP; }
    g_den_persist2_state_pd.ref() = ok ? 1 : 0;
    if (!ok) return PK2_OK;
  }
  sc.ntasks = p.ntasks;
  *ran = true;
  if (with_num) *num_ran = true;
  return PK2_OK;
}

void den_persist2_check_launch(float* den_lp, int N, hipStream_t stream) {
  DenPersist2Scratch& sc = g_den2_scratch[dev_stream(stream)];
  PersistGuard guard;
  (void)persist_guard(&guard);
  if (sc.ctl && sc.ntasks > 0) {
    hipLaunchKernelGGL(den_persist2_check, dim3(1), dim3(256), 0, stream, sc.ctl, sc.ntasks, den_lp, N, guard.dev, guard.host_dev);
    sc.ctl_clean = true;
  }
}

bool den_persist2_tail_check(hipStream_t stream, DenTailCheck* ck) {
  DenPersist2Scratch& sc = g_den2_scratch[dev_stream(stream)];
  PersistGuard guard;
  (void)persist_guard(&guard);
  if (!sc.ctl || sc.ntasks <= 0) return false;
  ck->ctl = reinterpret_cast<unsigned*>(sc.ctl);
  ck->ctl_words = (int)(sizeof(DenPersistCtl) / sizeof(unsigned));
  ck->ntasks = sc.ntasks;
  ck->abort_word = &sc.ctl->abort; ck->done_word = &sc.ctl->done; ck->count_word = &sc.ctl->pad[0];
  ck->guard_dev = guard.dev; ck->guard_host = guard.host_dev;
  sc.ctl_clean = true;
  return true;
}

}  // namespace pk2
