// Denominator graph of the LF-MMI objective: host-side construction and C ABI (gfx950 library).
//
// Replaces kaldi.chain.DenominatorGraph (reference bin/train_chain.py:167,202): reads den.fst
// (OpenFst binary) or arc arrays, computes initial_probs (Kaldi's 100-iteration rule, SURVEY.md
// Appendix A.2) and lays the arcs out for the MI355X kernels of chain_den.hip: three orderings
// (by destination for alpha, by source for beta, by pdf for the occupancies), each cut into
// workgroup chunks of whole rows and stored lane-interleaved (chain_internal.h).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "chain_internal.h"
#include "den_persist.h"
#include "openfst_io.h"

namespace pk2 {

// ----------------------------------------------------------------------------------------
// host: graph construction
// ----------------------------------------------------------------------------------------
// `group_of_row` (optional, monotone): rows of one group (the virtual states of one real state) are kept in ONE chunk so
// that the forward epilogue can sum them without atomics; a group that cannot fit a chunk is emitted row by row as
// single-row `atomic` chunks.  Without it every row is its own group (a row longer than a chunk is split, as before).
// `want4` / `want2`: which record formats to lay out (16-byte {a, b, prob, pi*prob} / 8-byte {a, prob}).
// `true_ends`: bit j of a lane's mask is set only where a row really ends after arc j (and at the chunk's last slot), not
// at every lane end: the state-x kernels carry the open tail of a lane to the next lanes with a wave scan instead of
// flushing it with an LDS atomic; wb_crow[wb] = chunk-local row that is open at the end of wave block wb.
static void build_ordering(int64_t A, int num_rows, const int32_t* key, const int32_t* a,
                           const int32_t* b, const float* prob, const float* piprob,
                           HostOrdering* out, const int32_t* group_of_row = nullptr,
                           bool want4 = true, bool want2 = true, bool true_ends = false) {
  // counting sort by key (stable)
  std::vector<int64_t> ptr(num_rows + 1, 0);
  for (int64_t i = 0; i < A; ++i) ptr[key[i] + 1]++;
  for (int r = 0; r < num_rows; ++r) ptr[r + 1] += ptr[r];
  std::vector<int64_t> perm(A);
  {
    std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
    for (int64_t i = 0; i < A; ++i) perm[cur[key[i]]++] = i;
  }
  out->arcs.clear(); out->meta.clear(); out->wb_off.assign(1, 0);
  out->arcs2.clear(); out->row_leak.clear(); out->slot0.clear();
  out->row0.clear(); out->nrows.clear(); out->atomic.clear();
  out->real0.clear(); out->nreal.clear(); out->wb_crow.clear();
  auto group = [&](int r) { return group_of_row ? group_of_row[r] : r; };

  struct Piece { int row; int64_t lo, hi; };  // arcs [lo,hi) of sorted list belong to `row`
  std::vector<int64_t> idx; std::vector<int> lrow;
  auto emit_chunk = [&](const std::vector<Piece>& pieces, int row0, int nrows, int atomic) {
    // sorted arcs of this chunk, each tagged with its chunk-local row
    idx.clear(); lrow.clear();
    for (const Piece& p : pieces) {
      for (int64_t k = p.lo; k < p.hi; ++k) { idx.push_back(perm[k]); lrow.push_back(p.row - row0); }
      // a row without arcs gets one null arc so that chunk-local rows stay consecutive
      if (p.lo == p.hi) { idx.push_back(-1); lrow.push_back(p.row - row0); }
    }
    const int per_wb = 64 * kK;
    int64_t n = (int64_t)idx.size();
    int64_t padded = std::max<int64_t>(per_wb, (n + per_wb - 1) / per_wb * per_wb);
    int last_row = nrows - 1;
    int nwb = (int)(padded / per_wb);
    const size_t base_arc = (size_t)out->wb_off.back() * per_wb, base_meta = out->meta.size();
    if (want4) out->arcs.resize(base_arc + padded);
    if (want2) out->arcs2.resize(base_arc + padded);
    // leaky-HMM term of a row: sum over its arcs of pi[src]*prob does not depend on the frame, so the state-x forward
    // kernel adds it per row instead of carrying pi*prob in every arc record
    const size_t base_slot = out->row_leak.size();
    out->slot0.push_back((int32_t)base_slot);
    {
      std::vector<double> leak(nrows, 0.0);
      for (size_t q = 0; q < idx.size(); ++q)
        if (idx[q] >= 0) leak[lrow[q]] += (double)piprob[idx[q]];
      for (int r = 0; r < nrows; ++r) out->row_leak.push_back((float)leak[r]);
    }
    out->meta.resize(base_meta + (size_t)nwb * 64);
    for (int wb = 0; wb < nwb; ++wb) {
      for (int lane = 0; lane < 64; ++lane) {
        uint32_t mask = 0; int c0 = 0;
        for (int j = 0; j < kK; ++j) {
          int64_t s = (int64_t)wb * per_wb + (int64_t)lane * kK + j;
          int4 rec; int row_here, row_next;
          if (s < n && idx[s] < 0) {
            rec.x = 0; rec.y = 0; rec.z = 0; rec.w = 0;
            row_here = lrow[s];
          } else if (s < n) {
            int64_t i = idx[s];
            rec.x = a[i]; rec.y = b[i];
            rec.z = __builtin_bit_cast(int, prob[i]);
            rec.w = __builtin_bit_cast(int, piprob[i]);
            row_here = lrow[s];
          } else {
            rec.x = 0; rec.y = 0; rec.z = 0; rec.w = 0;  // null arc: contributes exactly 0
            row_here = last_row;
          }
          row_next = (s + 1 < n) ? lrow[s + 1] : last_row;
          if (j == 0) c0 = row_here;
          const bool last_slot = s == padded - 1;
          if (true_ends ? (row_next != row_here || last_slot) : (j == kK - 1 || row_next != row_here)) mask |= (1u << j);
          if (lane == 63 && j == kK - 1) out->wb_crow.push_back(row_here);
          if (want4) out->arcs[base_arc + ((size_t)wb * kK + j) * 64 + lane] = rec;
          if (want2) out->arcs2[base_arc + ((size_t)wb * kK + j) * 64 + lane] = make_int2(rec.x, rec.z);
        }
        out->meta[base_meta + (size_t)wb * 64 + lane] = make_uint2((uint32_t)c0, mask);
      }
    }
    out->wb_off.push_back(out->wb_off.back() + nwb);
    out->row0.push_back(row0);
    out->nrows.push_back(nrows);
    out->atomic.push_back(atomic);
    out->real0.push_back(group(row0));
    out->nreal.push_back(group(row0 + nrows - 1) - group(row0) + 1);
  };

  std::vector<Piece> cur; int cur_row0 = 0; int64_t cur_arcs = 0;
  auto flush = [&](int next_row) {
    if (!cur.empty()) emit_chunk(cur, cur_row0, (int)cur.size(), 0);
    cur.clear(); cur_arcs = 0; cur_row0 = next_row;
  };
  for (int r = 0; r < num_rows;) {
    int r1 = r + 1;
    while (r1 < num_rows && group(r1) == group(r)) ++r1;
    int64_t total = 0, longest = 0;
    for (int q = r; q < r1; ++q) {
      const int64_t len = ptr[q + 1] - ptr[q];
      total += std::max<int64_t>(len, 1);  // empty rows carry one null arc
      longest = std::max(longest, len);
    }
    const bool grouped = r1 - r > 1;
    if (longest > kChunkArcs || (grouped && (total > kChunkArcs || r1 - r > kMaxRows))) {
      // does not fit a chunk: every row of the group in single-row atomic chunks of <= kChunkArcs arcs
      flush(r);
      bool first = true;   // the first piece of the row / group is flagged 2: per-state terms are added there, once
      for (int q = r; q < r1; ++q) {
        const int64_t lo = ptr[q], hi = ptr[q + 1];
        if (lo == hi && grouped) { emit_chunk({{q, lo, hi}}, q, 1, first ? 2 : 1); first = false; continue; }
        for (int64_t s = lo; s < hi; s += kChunkArcs) {
          emit_chunk({{q, s, std::min(hi, s + kChunkArcs)}}, q, 1, first ? 2 : 1);
          first = false;
        }
      }
      cur_row0 = r1;
    } else {
      if (cur_arcs + total > kChunkArcs || (int)cur.size() + (r1 - r) > kMaxRows) flush(r);
      for (int q = r; q < r1; ++q) cur.push_back({q, ptr[q], ptr[q + 1]});
      cur_arcs += total;
    }
    r = r1;
  }
  flush(num_rows);
  out->n_chunks = (int)out->row0.size();
}

// Persistent-kernel layout of one ordering (chain_internal.h: HostPersist).  `key` = row of an arc, `idx` = what it
// gathers, `group_of_row` (monotone, may be null: every row its own group) = the real state a row belongs to.
//  * `estep`: every row is padded with null slots to a multiple of estep slots, so a row can only end at the slots
//    j = estep-1 (mod estep) of a thread: the kernel runs its row-end code (an exec-masked LDS store and three VALU
//    operations) at 64/estep places per thread instead of 64.  Costs slots: the caller takes the largest step that fits.
//  * the arcs of a row that sit in one thread are dealt to its slots so that, for every slot index j, the 32 lanes of a
//    half wave gather from as many different LDS banks as possible (ds_read_b32: a half wave per cycle, bank = index
//    mod 32; random indices cost ~3.5 cycles per half wave).
static bool build_persist_step(int64_t A, int num_rows, const int32_t* key, const int32_t* idx, const float* prob,
                               const float* piprob, const int32_t* group_of_row, int num_groups, int estep, HostPersist* out) {
  *out = HostPersist();
  out->estep = estep;
  std::vector<int64_t> ptr(num_rows + 1, 0);
  for (int64_t i = 0; i < A; ++i) ptr[key[i] + 1]++;
  for (int r = 0; r < num_rows; ++r) ptr[r + 1] += ptr[r];
  std::vector<int64_t> perm(A);
  {
    std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
    for (int64_t i = 0; i < A; ++i) perm[cur[key[i]]++] = i;
  }
  auto slots_of = [&](int r) { return (std::max<int64_t>(1, ptr[r + 1] - ptr[r]) + estep - 1) / estep * estep; };
  // rows of each group
  std::vector<int32_t> grow(num_groups + 1, 0);
  if (group_of_row) {
    for (int r = 0; r < num_rows; ++r) grow[group_of_row[r] + 1]++;
    for (int g = 0; g < num_groups; ++g) grow[g + 1] += grow[g];
  } else {
    for (int g = 0; g <= num_groups; ++g) grow[g] = g;
  }
  if (grow[num_groups] != num_rows) return false;
  std::vector<int64_t> gslots(num_groups, 0);
  int64_t total = 0;
  for (int g = 0; g < num_groups; ++g) {
    for (int r = grow[g]; r < grow[g + 1]; ++r) gslots[g] += slots_of(r);
    total += gslots[g];
  }
  // contiguous ranges of whole groups, balanced by slots
  out->row_begin.assign(kPR + 1, num_rows);
  out->grp_begin.assign(kPR + 1, num_groups);
  int g = 0;
  int64_t left = total;
  for (int r = 0; r < kPR; ++r) {
    out->grp_begin[r] = g; out->row_begin[r] = grow[g];
    const int64_t target = (left + (kPR - r) - 1) / (kPR - r);
    int64_t have = 0; int rows = 0, groups = 0;
    while (g < num_groups) {
      const int gr = grow[g + 1] - grow[g];
      if (have + gslots[g] > kPSlots || rows + gr > kPMaxRows || groups + 1 > kPMaxRows) break;
      if (have >= target) break;
      have += gslots[g]; rows += gr; ++groups; ++g;
    }
    left -= have;
    out->max_rows = std::max(out->max_rows, rows);
    out->max_groups = std::max(out->max_groups, groups);
  }
  if (g < num_groups) return false;       // does not fit
  out->arcs.assign((size_t)kPR * kPSlots, make_int2(0, 0));
  out->ends.assign((size_t)kPR * kPT, 0ull);
  out->first_row.assign((size_t)kPR * kPT, 0);
  out->wcrow.assign((size_t)kPR * kPW, -1);
  out->row_leak.assign(num_rows, 0.f);
  out->row_psum.assign(num_rows, 0.f);
  std::vector<int32_t> srow; std::vector<int64_t> sarc; std::vector<char> send;
  std::vector<int32_t> sidx;              // gathered index of a slot after the bank-aware deal (nulls get one too)
  std::vector<int32_t> load((size_t)2 * kPK * 32);
  std::vector<int64_t> cand;
  for (int r = 0; r < kPR; ++r) {
    const int row0 = out->row_begin[r], row1 = out->row_begin[r + 1];
    srow.clear(); sarc.clear(); send.clear();
    for (int q = row0; q < row1; ++q) {
      double leak = 0.0, psum = 0.0;
      for (int64_t k = ptr[q]; k < ptr[q + 1]; ++k) {
        srow.push_back(q - row0); sarc.push_back(perm[k]); send.push_back(0);
        leak += (double)piprob[perm[k]]; psum += (double)prob[perm[k]];
      }
      if (ptr[q] == ptr[q + 1]) { srow.push_back(q - row0); sarc.push_back(-1); send.push_back(0); }
      while (srow.size() % estep) { srow.push_back(q - row0); sarc.push_back(-1); send.push_back(0); }
      send.back() = 1;
      out->row_leak[q] = (float)leak;
      out->row_psum[q] = (float)psum;
    }
    const int64_t n = (int64_t)srow.size();
    // bank-aware deal: inside a thread, the slots of one row are interchangeable
    sidx.assign(n, 0);
    for (int64_t wbase = 0; wbase < n; wbase += (int64_t)64 * kPK) {
      std::fill(load.begin(), load.end(), 0);
      for (int lane = 0; lane < 64; ++lane) {
        const int half = lane >> 5;
        int64_t s = wbase + (int64_t)lane * kPK;
        const int64_t lane_end = std::min<int64_t>(n, s + kPK);
        while (s < lane_end) {
          int64_t e = s;
          while (e < lane_end && srow[e] == srow[s]) ++e;
          cand.assign(sarc.begin() + s, sarc.begin() + e);      // arcs (and nulls, -1) of this segment
          for (int64_t pos = s; pos < e; ++pos) {
            const int j = (int)(pos % kPK);
            int32_t* ld = &load[((size_t)half * kPK + j) * 32];
            size_t best = 0; int best_load = 1 << 30;
            for (size_t c = 0; c < cand.size(); ++c) {
              const int l = cand[c] < 0 ? -1 : ld[idx[cand[c]] & 31];     // a null slot goes wherever it is free
              if (l < best_load) { best_load = l; best = c; }
            }
            const int64_t a = cand[best];
            cand[best] = cand.back(); cand.pop_back();
            sarc[pos] = a;
            if (a >= 0) {
              sidx[pos] = idx[a];
            } else {
              int bank = 0;
              for (int b2 = 1; b2 < 32; ++b2) if (ld[b2] < ld[bank]) bank = b2;
              sidx[pos] = bank;               // any valid entry: its probability is 0
            }
            ld[sidx[pos] & 31]++;
          }
          s = e;
        }
      }
    }
    for (int tid = 0; tid < kPT; ++tid) {
      uint64_t e = 0;
      for (int j = 0; j < kPK; ++j) {
        const int64_t s = (int64_t)tid * kPK + j;
        if (s >= n) break;
        out->arcs[((size_t)r * kPK + j) * kPT + tid] = make_int2(sidx[s], sarc[s] >= 0 ? __builtin_bit_cast(int, prob[sarc[s]]) : 0);
        if (send[s]) e |= 1ull << j;
      }
      out->ends[(size_t)r * kPT + tid] = e;
      const int64_t s0 = (int64_t)tid * kPK;
      out->first_row[(size_t)r * kPT + tid] = s0 < n ? srow[s0] : 0;
    }
    for (int w = 0; w < kPW; ++w) {
      const int64_t last = (int64_t)(w + 1) * 64 * kPK - 1;
      if (last < n && !send[last]) out->wcrow[(size_t)r * kPW + w] = srow[last];
    }
  }
  // device form: probabilities and 16-bit indices apart
  int max_idx = 0;
  for (const int2& a : out->arcs) max_idx = std::max(max_idx, a.x);
  if (max_idx >= 65536) return false;
  out->prob.resize(out->arcs.size());
  out->idx2.assign(out->arcs.size() / 2, 0u);
  for (int r = 0; r < kPR; ++r)
    for (int j = 0; j < kPK; ++j)
      for (int tid = 0; tid < kPT; ++tid) {
        const int2 a = out->arcs[((size_t)r * kPK + j) * kPT + tid];
        out->prob[((size_t)r * kPK + j) * kPT + tid] = __builtin_bit_cast(float, a.y);
        out->idx2[((size_t)r * (kPK / 2) + j / 2) * kPT + tid] |= (uint32_t)a.x << (16 * (j & 1));
      }
  out->ok = true;
  return true;
}

static void build_persist(int64_t A, int num_rows, const int32_t* key, const int32_t* idx, const float* prob,
                          const float* piprob, const int32_t* group_of_row, int num_groups, HostPersist* out) {
  const char* env = getenv("PK2_DEN_ESTEP");
  const int forced = env ? atoi(env) : 0;
  for (int estep : {8, 4, 2, 1}) {
    if (forced > 0 && estep != forced) continue;
    if (build_persist_step(A, num_rows, key, idx, prob, piprob, group_of_row, num_groups, estep, out)) return;
  }
  *out = HostPersist();
}

// ---- second persistent layout (chain_internal.h: HostPersist2; kernel: chain_den_persist2.hip) ----------------------------
// Contiguous ranges of whole groups for the kPR ranks, balanced by arcs (+1 per row: every row costs a slot in every list).
static bool persist2_assign(int64_t A, int num_rows, const int32_t* key, const int32_t* group_of_row, int num_groups,
                            int max_rows_per_rank, HostPersist2* out, bool by_rows = false) {
  std::vector<int64_t> rarcs(num_rows, 0);
  for (int64_t i = 0; i < A; ++i) rarcs[key[i]]++;
  std::vector<int32_t> grow(num_groups + 1, 0);
  if (group_of_row) {
    for (int r = 0; r < num_rows; ++r) grow[group_of_row[r] + 1]++;
    for (int g = 0; g < num_groups; ++g) grow[g + 1] += grow[g];
  } else {
    for (int g = 0; g <= num_groups; ++g) grow[g] = g;
  }
  if (grow[num_groups] != num_rows) return false;
  // What a row costs next to its arcs.  With the slots nearly full (the bench graph: 1.01 M arcs on 1.05 M slots) the ranks
  // must be balanced by slots: 2 per row (padding).  A graph with FEWER arcs has slots to spare, and balancing it by arcs
  // alone makes the ranks' ROW counts uneven -- the LDS row arrays are sized by the largest rank, and every float they take
  // comes out of the state table: S = 55 k states with 0.5 M arcs got 1890 rows on its largest rank (mean 1719), 8 row
  // arrays of 1892 floats, a table 200 floats too small for the two-chunk form and fell to five chunks + 163 streamed pieces,
  // i.e. to the launch-per-frame kernels (18.1 us per frame), while the same states with TWICE the arcs kept the persistent
  // form (13.6; profiles/r04_den_sweep.txt).  The spare slots are spread over the rows as weight.
  // (only where the row arrays compete with the table: more than 2 kPT rows per rank on average.  Below that the heaviest
  // rank's arcs matter more -- its slack decides how widely rows can be padded: S = 10 k / A = 0.5 M lost its estep to the
  // weighting, 5.3 -> 6.1 us per frame, profiles/r05_den_sweep.txt mid-round)
  const int64_t spare = (int64_t)(0.92 * (double)kPR * (double)kPSlots) - A - 2 * (int64_t)num_rows;
  const int64_t row_w = num_rows <= 2 * kPT * kPR ? 2 : 2 + std::min<int64_t>(62, std::max<int64_t>(0, spare / std::max(1, num_rows)));
  // by_rows (round 6): the ranks get equal ROW counts whatever their arcs -- the last resort of a graph whose mean is within a
  // few rows of the most a workgroup's epilogues handle (S = 65 000: 2031 of 2048), where the greedy deal by cost leaves the
  // last rank more rows than that; the ranks' uneven arc counts then go to the streamed pieces.
  std::vector<int64_t> gcost(num_groups, 0);
  int64_t total = 0;
  for (int g = 0; g < num_groups; ++g) {
    if (by_rows) gcost[g] = grow[g + 1] - grow[g];
    else for (int r = grow[g]; r < grow[g + 1]; ++r) gcost[g] += rarcs[r] + row_w;
    total += gcost[g];
  }
  out->row_begin.assign(kPR + 1, num_rows);
  out->grp_begin.assign(kPR + 1, num_groups);
  out->max_rows = out->max_groups = 0;
  int g = 0;
  int64_t left = total;
  for (int r = 0; r < kPR; ++r) {
    out->grp_begin[r] = g; out->row_begin[r] = grow[g];
    const int64_t target = (left + (kPR - r) - 1) / (kPR - r);
    int64_t have = 0; int rows = 0, groups = 0;
    while (g < num_groups) {
      const int gr = grow[g + 1] - grow[g];
      if (rows + gr > max_rows_per_rank || groups + 1 > max_rows_per_rank) break;
      if (have >= target) break;
      have += gcost[g]; rows += gr; ++groups; ++g;
    }
    left -= have;
    out->max_rows = std::max(out->max_rows, rows);
    out->max_groups = std::max(out->max_groups, groups);
  }
  return g == num_groups;
}

// Table chunks of an ordering whose vector has R entries, for an LDS table of `tcap` floats (a multiple of 512).
static bool persist2_chunks(int64_t A, const int32_t* idx, int R, int num_rows, int tcap, HostPersist2* out) {
  out->R = R;
  const int rpad = (R + 255) / 256 * 256;
  if (rpad <= tcap) {
    // Back to back.  Pass A may only gather from chunk 0 (chunk 1 is copied while it runs), pass B from both, so arcs into
    // chunk 0 can go to either list.  Chunk 0's copy is exposed, chunk 1's hides behind pass A: chunk 1 as large as pass A can
    // carry (64 rows: one copy instruction per wave at each of its 8 issue places), chunk 0 at least large enough that list
    // B is left with no more than its register slots (0.94 of them: rows are padded) -- and at least one row.
    const int trows = rpad / 256;
    std::vector<int64_t> cnt(trows, 0);
    for (int64_t i = 0; i < A; ++i) cnt[idx[i] / 256]++;
    const double need = std::max(0.02, 1.0 - 0.94 * (double)kPR * kQ * kPT / ((double)A + 0.5 * num_rows));
    int b = 0; int64_t below = 0;
    while (b < trows && (double)below < need * (double)A) below += cnt[b++];
    b = std::max({1, b, trows - 64});
    out->K = 2;
    out->cbeg[0] = 0; out->cbeg[1] = std::min(R, b * 256); out->cbeg[2] = R;
    out->lds_off[0] = 0; out->lds_off[1] = b * 256;
    out->tfloats = rpad;
    out->flexible = true;
    for (int c = 3; c <= kMaxChunks; ++c) out->cbeg[c] = R;
    return true;
  }
  out->flexible = false;
  const char* shared_env = getenv("PK2_DP2_SHARED");          // (read per graph: tests switch it)
  const bool shared_ok = !(shared_env && atoi(shared_env) == 0);
  if (shared_ok && rpad <= 2 * (tcap / 256 * 256)) {
    // Round 4: a vector of up to TWICE the LDS table keeps both resident passes.  Two chunks that take turns in ONE buffer:
    // pass A gathers from chunk 0, the buffer is then overwritten with chunk 1 (nothing hides that copy), pass B gathers
    // from chunk 1 only -- the lists follow the chunks strictly, so the split is put where half of the arcs gather from
    // below it (both halves must fit the buffer).  Every arc stays in registers: S = 30 k -> 50 k states at 1.0 M arcs
    // costs the second, exposed table copy (+2 us per frame) instead of the launch-per-frame kernels (x 2.9).
    const int trows = rpad / 256, brows = tcap / 256;
    std::vector<int64_t> cnt(trows, 0);
    for (int64_t i = 0; i < A; ++i) cnt[idx[i] / 256]++;
    int b = 0; int64_t below = 0;
    while (b < trows && 2 * below < A) below += cnt[b++];
    b = std::max(trows - brows, std::min(b, brows));
    b = std::max(1, std::min(b, trows - 1));
    out->K = 2;
    out->cbeg[0] = 0; out->cbeg[1] = std::min(R, b * 256); out->cbeg[2] = R;
    out->lds_off[0] = 0; out->lds_off[1] = 0;
    out->tfloats = std::max(b, trows - b) * 256;
    for (int c = 3; c <= kMaxChunks; ++c) out->cbeg[c] = R;
    return true;
  }
  const int H = tcap / 2 / 256 * 256;
  if (H < 256) return false;
  const int K = (R + H - 1) / H;
  if (K > kMaxChunks) return false;
  out->K = K;
  for (int c = 0; c <= kMaxChunks; ++c) out->cbeg[c] = std::min(R, c * H);
  for (int c = 0; c < K; ++c) out->lds_off[c] = (c & 1) * H;
  out->tfloats = 2 * H;
  return true;
}

// Deals the arcs of a row that sit in one thread to its slots so that, slot index by slot index, the 32 lanes of a half wave
// gather from different LDS banks (as build_persist_step).  `spt` = consecutive sorted slots per thread.
static void persist2_deal(int64_t n, int spt, const std::vector<int32_t>& srow, std::vector<int64_t>& sarc, const int32_t* lidx,
                          std::vector<int32_t>& sidx, int null_base, int null_span) {
  sidx.assign(n, null_base);
  std::vector<int32_t> load((size_t)2 * spt * 32);
  std::vector<int64_t> cand;
  for (int64_t wbase = 0; wbase < n; wbase += (int64_t)64 * spt) {
    std::fill(load.begin(), load.end(), 0);
    for (int lane = 0; lane < 64; ++lane) {
      const int half = lane >> 5;
      int64_t s = wbase + (int64_t)lane * spt;
      const int64_t lane_end = std::min<int64_t>(n, s + spt);
      while (s < lane_end) {
        int64_t e = s;
        while (e < lane_end && srow[e] == srow[s]) ++e;
        cand.assign(sarc.begin() + s, sarc.begin() + e);
        for (int64_t pos = s; pos < e; ++pos) {
          const int j = (int)(pos % spt);
          int32_t* ld = &load[((size_t)half * spt + j) * 32];
          size_t best = 0; int best_load = 1 << 30;
          for (size_t c = 0; c < cand.size(); ++c) {
            const int l = cand[c] < 0 ? -1 : ld[lidx[cand[c]] & 31];
            if (l < best_load) { best_load = l; best = c; }
          }
          const int64_t a = cand[best];
          cand[best] = cand.back(); cand.pop_back();
          sarc[pos] = a;
          if (a >= 0) {
            sidx[pos] = lidx[a];
          } else {
            int bank = null_base & 31;          // any VALID entry of the chunk's buffer (its probability is 0; what lies
            for (int b2 = 0; b2 < null_span; ++b2)   // past the vector's end in LDS was never written and may be a NaN)
              if (ld[(null_base + b2) & 31] < ld[bank]) bank = (null_base + b2) & 31;
            int off = 0;
            while (((null_base + off) & 31) != bank) ++off;
            sidx[pos] = null_base + off;
          }
          ld[sidx[pos] & 31]++;
        }
        s = e;
      }
    }
  }
}

// Bank-aware deal of a resident pass, row by row: ALL slots of a row (they may lie in several lanes, even waves) are
// interchangeable, and lanes of a half wave that read the SAME address are served by one broadcast -- what a 32-lane
// ds_read_b32 costs is the largest number of DISTINCT addresses in one of the 32 banks.  Greedy over the rows in order
// (slot by slot the arc whose address is already read at that (half wave, slot index), else the one whose bank holds the
// fewest addresses there), then `rounds` passes in which every row is taken out and dealt again against all others.
// On the BASELINE graph: 2.85 -> see tools/dbg/den_bank_conflicts.py cycles per half-wave gather (random: 3.53).
static void persist2_deal_rows(int64_t n, int spt, const std::vector<int32_t>& srow, std::vector<int64_t>& sarc, const int32_t* lidx,
                               std::vector<int32_t>& sidx, int null_base, int null_span, int rounds) {
  sidx.assign(n, null_base);
  if (n == 0) return;
  if (rounds < 0) {                       // (a sizing run of the layout: any valid assignment)
    for (int64_t i = 0; i < n; ++i) sidx[i] = sarc[i] >= 0 ? lidx[sarc[i]] : null_base;
    return;
  }
  const int nhw = (int)((n + (int64_t)32 * spt - 1) / ((int64_t)32 * spt));
  // per (half wave hw, slot index j): the <= 32 distinct addresses its lanes read, with the number of lanes reading each,
  // and cnt[bank] = distinct addresses in that bank
  struct Cell { int32_t addr[32]; uint8_t refs[32]; uint8_t cnt[32]; int n; };
  std::vector<Cell> cells((size_t)nhw * spt);
  for (Cell& c : cells) { c.n = 0; for (int b = 0; b < 32; ++b) c.cnt[b] = 0; }
  auto cell_of = [&](int64_t pos) -> Cell& { return cells[(size_t)(pos / ((int64_t)32 * spt)) * spt + (size_t)(pos % spt)]; };
  auto put = [&](int64_t pos, int addr) {
    Cell& c = cell_of(pos);
    for (int k = 0; k < c.n; ++k) if (c.addr[k] == addr) { c.refs[k]++; return; }
    c.addr[c.n] = addr; c.refs[c.n] = 1; c.n++; c.cnt[addr & 31]++;
  };
  auto take = [&](int64_t pos, int addr) {
    Cell& c = cell_of(pos);
    for (int k = 0; k < c.n; ++k)
      if (c.addr[k] == addr) {
        if (--c.refs[k] == 0) { c.cnt[addr & 31]--; c.n--; c.addr[k] = c.addr[c.n]; c.refs[k] = c.refs[c.n]; }
        return;
      }
  };
  auto cost = [&](int64_t pos, int addr) {
    const Cell& c = cell_of(pos);
    for (int k = 0; k < c.n; ++k) if (c.addr[k] == addr) return -1;      // a broadcast: free
    return (int)c.cnt[addr & 31];
  };
  auto null_addr = [&](int64_t pos) {
    int best = null_base, best_c = 1 << 30;
    for (int b2 = 0; b2 < null_span; ++b2) { const int c = cost(pos, null_base + b2); if (c < best_c) { best_c = c; best = null_base + b2; } }
    return best;
  };
  std::vector<int64_t> cand;
  auto deal_row = [&](int64_t s, int64_t e) {
    cand.assign(sarc.begin() + s, sarc.begin() + e);
    for (int64_t pos = s; pos < e; ++pos) {
      size_t best = 0; int best_c = 1 << 30;
      // (a very long row: the first 256 candidates are choice enough)
      const size_t lim = std::min<size_t>(cand.size(), 256);
      for (size_t c = 0; c < lim; ++c) {
        const int cc = cand[c] < 0 ? 1 << 20 : cost(pos, lidx[cand[c]]);      // nulls last: they can go anywhere
        if (cc < best_c) { best_c = cc; best = c; }
      }
      const int64_t a = cand[best];
      cand[best] = cand.back(); cand.pop_back();
      sarc[pos] = a;
      sidx[pos] = a >= 0 ? lidx[a] : null_addr(pos);
      put(pos, sidx[pos]);
    }
  };
  // first deal
  for (int64_t s = 0; s < n;) {
    int64_t e = s;
    while (e < n && srow[e] == srow[s]) ++e;
    deal_row(s, e);
    s = e;
  }
  for (int r = 0; r < rounds; ++r)
    for (int64_t s = 0; s < n;) {
      int64_t e = s;
      while (e < n && srow[e] == srow[s]) ++e;
      for (int64_t pos = s; pos < e; ++pos) take(pos, sidx[pos]);
      deal_row(s, e);
      s = e;
    }
}

// Which list every arc goes to (and the LDS offset of the entry it gathers): the chunk it gathers from -- except that with
// back-to-back chunks (`flexible`) pass B sees chunk 0 as well, so arcs into chunk 0 are moved to list 1 until the two lists
// of a rank are equally long.  Row by row, the split is taken from the few values around the proportional share that cost
// the fewest padding slots (both parts a multiple of `estep`, or one of them empty).
static bool persist2_arc_lists(int64_t A, const int32_t* idx, const HostPersist2& h, int estep, const std::vector<int64_t>& ptr,
                               const std::vector<int64_t>& perm, std::vector<uint8_t>* chunk_out, std::vector<int32_t>* lidx_out) {
  const int K = h.K;
  std::vector<uint8_t>& chunk = *chunk_out; std::vector<int32_t>& lidx = *lidx_out;
  chunk.assign(A, 0); lidx.assign(A, 0);
  for (int64_t i = 0; i < A; ++i) {
    int c = 0;
    while (c + 1 < K && idx[i] >= h.cbeg[c + 1]) ++c;
    chunk[i] = (uint8_t)c;
    lidx[i] = h.lds_off[c] + (idx[i] - h.cbeg[c]);
    if (lidx[i] >= 65536) return false;
  }
  if (!h.flexible) return true;
  auto pad = [&](int64_t x) { return x == 0 ? 0 : (estep - x % estep) % estep; };
  for (int r = 0; r < kPR; ++r) {
    int64_t elig = 0, all = 0;
    for (int q = h.row_begin[r]; q < h.row_begin[r + 1]; ++q)
      for (int64_t k = ptr[q]; k < ptr[q + 1]; ++k) { ++all; if (chunk[perm[k]] == 0) ++elig; }
    if (elig == 0) continue;
    const double keep = std::min(1.0, (double)all / 2.0 / (double)elig);   // share of the eligible arcs that stays in list 0
    double err = 0.0;
    for (int q = h.row_begin[r]; q < h.row_begin[r + 1]; ++q) {
      int64_t e = 0, o = 0;
      for (int64_t k = ptr[q]; k < ptr[q + 1]; ++k) (chunk[perm[k]] == 0 ? e : o)++;
      if (e == 0) continue;
      const double t = keep * (double)e + err;
      int64_t best = -1; double best_cost = 1e30;
      const int64_t lo = std::max<int64_t>(0, (int64_t)std::floor(t) - estep), hi = std::min<int64_t>(e, (int64_t)std::ceil(t) + estep);
      for (int64_t a = lo; a <= hi; ++a) {
        const double cost = (double)(pad(a) + pad(e - a + o)) + 0.01 * std::fabs((double)a - t);
        if (cost < best_cost) { best_cost = cost; best = a; }
      }
      err = t - (double)best;
      int64_t kept = 0;
      for (int64_t k = ptr[q]; k < ptr[q + 1]; ++k) {
        if (chunk[perm[k]] != 0) continue;
        if (kept < best) ++kept; else chunk[perm[k]] = 1;
      }
    }
  }
  return true;
}

// The slot lists of an ordering whose ranks, chunks and arc lists are fixed.  `res` = register slots per thread of a resident
// pass (kQ; smaller values exist for tests that force streaming on small graphs).
//  * resident list c (c < 2): only the rows that have an arc in it, numbered compactly (rmap: rank-local compact index of a
//    row in list c, -1 = absent); what exceeds res * kPT slots is cut off and streamed;
//  * streamed segment c: the cut-off remainder of list c (from the row the cut falls into) or, for c >= 2, the whole list,
//    with EVERY row from its first one on (a null slot where a row has nothing): it adds into a row-indexed array.
static bool persist2_lists(int64_t A, int num_rows, const float* prob, const float* piprob, const std::vector<int64_t>& ptr,
                           const std::vector<int64_t>& perm, const std::vector<uint8_t>& chunk, const std::vector<int32_t>& lidx,
                           int estep, int res, bool deal, HostPersist2* out) {
  out->estep = estep;
  const int K = out->K;
  out->prob.assign((size_t)kPR * kPSlots, 0.f);
  out->idx2.assign((size_t)kPR * kPSlots / 2, 0u);
  out->ends.assign((size_t)kPR * 2 * kPT, 0u);
  out->first_row.assign((size_t)kPR * 2 * kPT, 0);
  out->uncovered.assign((size_t)kPR * 2, 0);
  out->ncomp.assign((size_t)kPR * 2, 0);
  out->rmap.assign((size_t)2 * num_rows, (int16_t)-1);
  out->pbeg.assign((size_t)kPR * (kMaxChunks + 1), 0);
  out->sfirst_row.assign((size_t)kPR * kMaxChunks * kPT, 0);
  out->wcrow.assign((size_t)kPR * kSegs * kPW, -1);
  out->row_leak.assign(num_rows, 0.f);
  out->row_psum.assign(num_rows, 0.f);
  out->sprob.clear(); out->sidx2.clear(); out->sends.clear();
  out->resident_slots = out->streamed_slots = 0;
  out->max_pieces = 0;
  std::vector<int32_t> srow, scomp, sidx; std::vector<int64_t> sarc; std::vector<char> send;
  int npieces = 0;
  // where the null slots of a chunk's lists point: valid entries of the buffer the chunk occupies (an empty chunk: chunk 0's)
  int null_base[kMaxChunks], null_span[kMaxChunks];
  for (int c = 0; c < K; ++c) {
    const int cc = out->cbeg[c + 1] > out->cbeg[c] ? c : 0;
    null_base[c] = out->lds_off[cc];
    null_span[c] = std::max(1, std::min(32, out->cbeg[cc + 1] - out->cbeg[cc]));
  }
  for (int r = 0; r < kPR; ++r) {
    const int row0 = out->row_begin[r], row1 = out->row_begin[r + 1];
    for (int q = row0; q < row1; ++q) {
      double leak = 0.0, psum = 0.0;
      for (int64_t k = ptr[q]; k < ptr[q + 1]; ++k) { leak += (double)piprob[perm[k]]; psum += (double)prob[perm[k]]; }
      out->row_leak[q] = (float)leak; out->row_psum[q] = (float)psum;
    }
    int rank_pieces = 0;
    for (int c = 0; c < K; ++c) {
      // the list of chunk c: the rows with an arc in it, padded to estep slots; scomp = compact row index
      srow.clear(); scomp.clear(); sarc.clear(); send.clear();
      int ncomp = 0;
      for (int q = row0; q < row1; ++q) {
        const size_t before = srow.size();
        for (int64_t k = ptr[q]; k < ptr[q + 1]; ++k)
          if (chunk[perm[k]] == c) { srow.push_back(q - row0); scomp.push_back(ncomp); sarc.push_back(perm[k]); send.push_back(0); }
        if (srow.size() == before) continue;
        while ((srow.size() - before) % estep) { srow.push_back(q - row0); scomp.push_back(ncomp); sarc.push_back(-1); send.push_back(0); }
        send.back() = 1;
        if (c < 2) out->rmap[(size_t)c * num_rows + q] = (int16_t)ncomp;
        ++ncomp;
      }
      const int64_t n = (int64_t)srow.size();
      const int64_t nres = c < 2 ? std::min<int64_t>(n, (int64_t)res * kPT) : 0;
      // (a cut falls on a slot boundary of estep: res and kSP are multiples of it)
      if (c < 2) {
        // resident pass: thread tid owns the sorted slots tid*res .. tid*res + res-1
        std::vector<int32_t> rrow(scomp.begin(), scomp.begin() + nres);
        std::vector<int64_t> rarc(sarc.begin(), sarc.begin() + nres);
        persist2_deal_rows(nres, res, rrow, rarc, lidx.data(), sidx, null_base[c], null_span[c], deal ? 2 : -1);
        for (int tid = 0; tid < kPT; ++tid) {
          uint32_t e = 0;
          for (int j = 0; j < res; ++j) {
            const int64_t s = (int64_t)tid * res + j;
            if (s >= nres) break;
            const int jj = c * kQ + j;
            out->prob[((size_t)r * kPK + jj) * kPT + tid] = rarc[s] >= 0 ? prob[rarc[s]] : 0.f;
            out->idx2[((size_t)r * (kPK / 2) + jj / 2) * kPT + tid] |= (uint32_t)sidx[s] << (16 * (jj & 1));
            if (send[s]) e |= 1u << j;
          }
          // (slots past the list, and the unused slots res..kQ-1 of a test build, keep probability 0 and LDS offset 0: the
          // first entry of chunk 0 or of the chunk that replaced it in buffer 0, a valid number either way)
          out->ends[((size_t)r * 2 + c) * kPT + tid] = e;
          const int64_t s0 = (int64_t)tid * res;
          out->first_row[((size_t)r * 2 + c) * kPT + tid] = s0 < nres ? scomp[s0] : 0;
        }
        for (int w = 0; w < kPW; ++w) {      // (the carries are addressed by rank-local ROW, like the streamed segments')
          const int64_t last = (int64_t)(w + 1) * 64 * res - 1;
          if (last < nres && !send[last]) out->wcrow[((size_t)r * kSegs + c) * kPW + w] = srow[last];
        }
        // compact rows whose END is not in the resident part get no store from the pass: the kernel clears them itself
        out->uncovered[(size_t)r * 2 + c] = nres < n ? scomp[nres] : ncomp;
        out->ncomp[(size_t)r * 2 + c] = ncomp;
        out->resident_slots += nres;
      }
      // streamed segment: the remainder, every row from its first one on; thread tid owns pieces*kSP consecutive sorted slots
      out->pbeg[(size_t)r * (kMaxChunks + 1) + c] = npieces;
      if (nres == n) continue;
      {
        std::vector<int32_t> rrow; std::vector<int64_t> rarc; std::vector<char> rend;
        int64_t s = nres;
        for (int q = srow[nres]; q < row1 - row0; ++q) {
          const size_t before = rrow.size();
          while (s < n && srow[s] == q) { if (sarc[s] >= 0) { rrow.push_back(q); rarc.push_back(sarc[s]); rend.push_back(0); } ++s; }
          if (rrow.size() == before) { rrow.push_back(q); rarc.push_back(-1); rend.push_back(0); }
          while ((rrow.size() - before) % estep) { rrow.push_back(q); rarc.push_back(-1); rend.push_back(0); }
          rend.back() = 1;
        }
        const int64_t ns = (int64_t)rrow.size();
        const int pieces = (int)((ns + (int64_t)kPT * kSP - 1) / ((int64_t)kPT * kSP));
        const int spt = pieces * kSP;
        if (deal) persist2_deal(ns, spt, rrow, rarc, lidx.data(), sidx, null_base[c], null_span[c]);
        else { sidx.assign(ns, null_base[c]); for (int64_t i = 0; i < ns; ++i) if (rarc[i] >= 0) sidx[i] = lidx[rarc[i]]; }
        out->sprob.resize((size_t)(npieces + pieces) * kSP * kPT, 0.f);
        out->sidx2.resize((size_t)(npieces + pieces) * (kSP / 2) * kPT, 0u);
        out->sends.resize((size_t)(npieces + pieces) * kPT, 0u);
        // null slots past the segment's end gather a valid entry too
        const uint32_t null_pair = (uint32_t)null_base[c] | ((uint32_t)null_base[c] << 16);
        for (size_t k = (size_t)npieces * (kSP / 2) * kPT; k < out->sidx2.size(); ++k) out->sidx2[k] = null_pair;
        for (int tid = 0; tid < kPT; ++tid) {
          for (int p = 0; p < spt; ++p) {
            const int64_t s2 = (int64_t)tid * spt + p;
            if (s2 >= ns) break;
            const int piece = npieces + p / kSP, j = p % kSP;
            out->sprob[((size_t)piece * kPT + tid) * kSP + j] = rarc[s2] >= 0 ? prob[rarc[s2]] : 0.f;
            uint32_t& w2 = out->sidx2[((size_t)piece * kPT + tid) * (kSP / 2) + j / 2];
            w2 = (w2 & ~(0xffffu << (16 * (j & 1)))) | ((uint32_t)sidx[s2] << (16 * (j & 1)));
            if (rend[s2]) out->sends[(size_t)piece * kPT + tid] |= 1u << j;
          }
          const int64_t s0 = (int64_t)tid * spt;
          out->sfirst_row[((size_t)r * kMaxChunks + c) * kPT + tid] = s0 < ns ? rrow[s0] : 0;
        }
        for (int w = 0; w < kPW; ++w) {
          const int64_t last = (int64_t)(w + 1) * 64 * spt - 1;
          if (last < ns && !rend[last]) out->wcrow[((size_t)r * kSegs + 2 + c) * kPW + w] = rrow[last];
        }
        npieces += pieces;
        rank_pieces += pieces;
        out->streamed_slots += ns;
      }
    }
    for (int c = K; c <= kMaxChunks; ++c) out->pbeg[(size_t)r * (kMaxChunks + 1) + c] = npieces;
    out->max_pieces = std::max(out->max_pieces, rank_pieces);
  }
  out->ok = true;
  return true;
}

// Both orderings of the second persistent layout.  fkey / fidx: forward ordering (rows = virtual destination states,
// gathering source states); bkey / bidx: backward ordering (rows = source states, gathering virtual destination states).
static void build_persist2(pk2_den_graph* g, int64_t A2, const int32_t* arc_v, const int32_t* src2, const float* prob2,
                           const float* piprob2, const int32_t* vstate) {
  HostPersist2& f = g->h_p2fwd; HostPersist2& b = g->h_p2bwd;
  f = HostPersist2(); b = HostPersist2();
  g->p2_cap = 0;
  const int S = g->S, V = g->V;
  if (V >= 65536 || S >= 65536) { if (getenv("PK2_DP2_DEBUG")) fprintf(stderr, "persist2: bail at line %d\n", __LINE__); return; }
  // Rows per rank: at most 2 * kPT when that is possible (the row epilogues then handle two entries per thread and keep half
  // the per-state constants in registers; the passes cost the same however full their slots are), else up to kPMaxRows.
  bool assigned = false;
  for (int lim : {2 * kPT, 3 * kPT, kPMaxRows}) {
    if (!persist2_assign(A2, V, arc_v, vstate, S, lim, &f) || !persist2_assign(A2, S, src2, nullptr, S, lim, &b)) continue;
    // the backward workgroups also stage x for their own virtual states: max_groups = how many
    b.max_groups = 0;
    for (int r = 0; r < kPR; ++r) b.max_groups = std::max(b.max_groups, g->voff[b.row_begin[r + 1]] - g->voff[b.row_begin[r]]);
    if (b.max_groups > lim) continue;
    assigned = true;
    break;
  }
  if (!assigned && persist2_assign(A2, V, arc_v, vstate, S, kPMaxRows, &f, true) && persist2_assign(A2, S, src2, nullptr, S, kPMaxRows, &b, true)) {
    b.max_groups = 0;
    for (int r = 0; r < kPR; ++r) b.max_groups = std::max(b.max_groups, g->voff[b.row_begin[r + 1]] - g->voff[b.row_begin[r]]);
    assigned = b.max_groups <= kPMaxRows;
    if (assigned && getenv("PK2_DP2_DEBUG")) fprintf(stderr, "persist2: ranks dealt by rows (%d / %d rows on the largest)\n", f.max_rows, b.max_rows);
  }
  if (!assigned) { if (getenv("PK2_DP2_DEBUG")) fprintf(stderr, "persist2: bail at line %d\n", __LINE__); return; }
  const int cap = (std::max({f.max_rows, f.max_groups, b.max_rows, b.max_groups, 1}) + 3) / 4 * 4;
  // The eighth row array (pdfs: the kernel then gathers x from plain exp(logits) rows) takes `cap` floats from the table: kept
  // unless it costs either direction a table chunk (S = 55 k at 1.0 M arcs: 16.5 -> 22.8 us per frame with it).
  auto tcap_for = [&](int arrays) { return ((int64_t)kDenPersistMaxLds / 4 - arrays * (int64_t)cap - kP2FixedFloats) / 512 * 512; };
  g->p2_rowarrays = kP2RowArrays;
  if (tcap_for(kP2RowArrays) >= 512) {
    for (int which = 0; which < 2; ++which) {
      HostPersist2 with = which == 0 ? f : b, without = with;
      const int rows = which == 0 ? V : S, R = which == 0 ? S : V;
      const int32_t* idx = which == 0 ? src2 : arc_v;
      const bool ok8 = persist2_chunks(A2, idx, R, rows, (int)tcap_for(kP2RowArrays), &with);
      const bool ok7 = persist2_chunks(A2, idx, R, rows, (int)tcap_for(kP2RowArrays - 1), &without);
      if (ok7 && (!ok8 || without.K < with.K)) g->p2_rowarrays = kP2RowArrays - 1;
    }
  } else {
    g->p2_rowarrays = kP2RowArrays - 1;
  }
  if (const char* env = getenv("PK2_DP2_ROWARRAYS")) g->p2_rowarrays = atoi(env) == 7 ? 7 : 8;
  int64_t tcap = tcap_for(g->p2_rowarrays);
  if (const char* env = getenv("PK2_DP2_TCAP")) tcap = std::min<int64_t>(tcap, std::max(512, atoi(env) / 512 * 512));
  if (tcap < 512) { if (getenv("PK2_DP2_DEBUG")) fprintf(stderr, "persist2: bail at line %d\n", __LINE__); return; }
  int res = kQ;
  if (const char* env = getenv("PK2_DP2_RES")) res = std::max(1, std::min(kQ, atoi(env)));     // (tests: force streaming)
  const char* env_e = getenv("PK2_DEN_ESTEP");
  const int forced = env_e ? atoi(env_e) : 0;
  // per frame, in microseconds: what the arcs of the two resident passes cost with row-end code at 64 / estep places
  // (measured for 1 and 2 on the BASELINE graph), a streamed piece, a segment's extra barrier
  auto arcs_us = [](int e) { return e == 1 ? 3.1 : e == 2 ? 1.8 : e == 4 ? 1.55 : 1.45; };
  for (int which = 0; which < 2; ++which) {
    HostPersist2& h = which == 0 ? f : b;
    const int rows = which == 0 ? V : S, R = which == 0 ? S : V;
    const int32_t* key = which == 0 ? arc_v : src2;
    const int32_t* idx = which == 0 ? src2 : arc_v;
    if (!persist2_chunks(A2, idx, R, rows, (int)tcap, &h)) { if (getenv("PK2_DP2_DEBUG")) fprintf(stderr, "persist2: bail at line %d\n", __LINE__); f.ok = b.ok = false; return; }
    std::vector<int64_t> ptr(rows + 1, 0);
    for (int64_t i = 0; i < A2; ++i) ptr[key[i] + 1]++;
    for (int r = 0; r < rows; ++r) ptr[r + 1] += ptr[r];
    std::vector<int64_t> perm(A2);
    {
      std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
      for (int64_t i = 0; i < A2; ++i) perm[cur[key[i]]++] = i;
    }
    // the widest row padding without a streamed piece; if every width streams, the cheapest frame
    int best_estep = 0; double best_cost = 1e30;
    std::vector<uint8_t> list_of; std::vector<int32_t> lidx;
    for (int estep : {8, 4, 2, 1}) {        // sizing runs (no bank-aware deal)
      if ((forced > 0 && estep != forced) || res % estep != 0) continue;      // (a cut must fall between whole rows' slots)
      HostPersist2 cand = h;
      if (!persist2_arc_lists(A2, idx, cand, estep, ptr, perm, &list_of, &lidx) ||
          !persist2_lists(A2, rows, prob2, piprob2, ptr, perm, list_of, lidx, estep, res, false, &cand)) { if (getenv("PK2_DP2_DEBUG")) fprintf(stderr, "persist2: bail at line %d\n", __LINE__); f.ok = b.ok = false; return; }
      // (round 4: the FIRST streamed piece costs ~5 us per frame -- the streaming variant of the frame: zeroed row arrays,
      // spill reloads behind the piece loads -- and ~0.9 each after it (profiles/r04_den_sweep.txt: S = 30 k, 1.0 -> 1.5 M
      // arcs = 4 pieces: 7.75 -> 15.1); the old 0.3 + 0.6 per piece chose padded rows + one piece over unpadded rows)
      // (round 5: between two layouts that both stream the fixed price cancels; against a layout WITHOUT a piece it stays
      // -- with 0.4 + 0.9 per piece, the S = 30 k figures after the round's work on the pieces (profiles/r05_den_stream.txt),
      // S = 40 / 50 / 55 k at 1.0 M arcs chose padded rows + one piece and ran 10.8 / 13.9 / 14.9 us per frame instead of
      // 9.7 / 10.4 / 11.1 with unpadded rows and none: chunks sharing a buffer, or a third chunk, put full barriers between
      // the pieces in flight)
      const double cost = arcs_us(estep) + (cand.max_pieces ? 4.5 + 0.9 * cand.max_pieces : 0.0);
      if (cost < best_cost) { best_cost = cost; best_estep = estep; }
      if (cand.max_pieces == 0 && cand.K == 2) break;
    }
    if (best_estep == 0) { if (getenv("PK2_DP2_DEBUG")) fprintf(stderr, "persist2: bail at line %d\n", __LINE__); f.ok = b.ok = false; return; }
    if (!persist2_arc_lists(A2, idx, h, best_estep, ptr, perm, &list_of, &lidx) ||
        !persist2_lists(A2, rows, prob2, piprob2, ptr, perm, list_of, lidx, best_estep, res, true, &h)) { f.ok = b.ok = false; return; }
    if (getenv("PK2_DP2_DEBUG")) fprintf(stderr, "persist2: %s rows padded to %d slots, %d table chunks, at most %d streamed pieces per rank\n", which == 0 ? "fwd" : "bwd", best_estep, h.K, h.max_pieces);
  }
  if (!(f.ok && b.ok)) { if (getenv("PK2_DP2_DEBUG")) fprintf(stderr, "persist2: bail at line %d\n", __LINE__); f.ok = b.ok = false; return; }
  g->p2_cap = cap;
}

static int build_graph_ordered(int32_t S, int32_t P, int64_t A, const int32_t* src_in, const int32_t* dst_in,
                               const int32_t* pdf, const float* prob, int32_t start, const char* order_mode,
                               pk2_den_graph** out) {
  PK2_REQUIRE(S > 0 && P > 0 && A > 0 && start >= 0 && start < S, "den graph: bad sizes");
  PK2_REQUIRE(P <= 65536, "den graph: num_pdfs %d > 65536 unsupported", P);
  for (int64_t i = 0; i < A; ++i) {
    PK2_REQUIRE(src_in[i] >= 0 && src_in[i] < S && dst_in[i] >= 0 && dst_in[i] < S && pdf[i] >= 0 && pdf[i] < P,
                "den graph: arc %lld out of range", (long long)i);
  }
  auto* g = new pk2_den_graph();
  g->S = S; g->P = P; g->A = A;
  // Internal state numbering.  States only index the alpha / beta tables (occupancies are per pdf), so the library is
  // free to renumber them: the states that most arcs enter come first, so that the hot part of the backward gather table
  // shares cache lines (denominator call 11.05 -> 10.77 ms on the bench graph, whose in-degrees are skewed like a phone
  // LM's; ordering by in + out degree is slower, 12.4 ms).  PK2_DEN_ORDER=none keeps the caller's numbering.
  std::vector<int32_t> new_of(S), src_v(A), dst_v(A);
  {
    std::vector<int32_t> order(S);
    std::iota(order.begin(), order.end(), 0);
    const char* env = order_mode;
    if (!(env && strcmp(env, "none") == 0)) {
      std::vector<int64_t> deg(S, 0);
      const bool both = env && strcmp(env, "degree") == 0;
      for (int64_t i = 0; i < A; ++i) { deg[dst_in[i]]++; if (both) deg[src_in[i]]++; }
      std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return deg[a] > deg[b]; });
    }
    for (int32_t k = 0; k < S; ++k) new_of[order[k]] = k;
    g->orig_of = order;
    for (int64_t i = 0; i < A; ++i) { src_v[i] = new_of[src_in[i]]; dst_v[i] = new_of[dst_in[i]]; }
    start = new_of[start];
  }
  const int32_t* src = src_v.data();
  const int32_t* dst = dst_v.data();
  g->start = start;
  // initial_probs: Kaldi DenominatorGraph::SetInitialProbs: 100 iterations
  // of the normalised forward recursion from the start state, averaged (double precision).
  {
    std::vector<double> cur(S, 0.0), nxt(S), avg(S, 0.0);
    cur[start] = 1.0;
    const int iters = 100;
    for (int it = 0; it < iters; ++it) {
      for (int s = 0; s < S; ++s) avg[s] += cur[s] / iters;
      std::fill(nxt.begin(), nxt.end(), 0.0);
      for (int64_t i = 0; i < A; ++i) nxt[dst[i]] += cur[src[i]] * (double)prob[i];
      double tot = 0.0;
      for (int s = 0; s < S; ++s) tot += nxt[s];
      for (int s = 0; s < S; ++s) cur[s] = nxt[s] / tot;
    }
    g->pi.resize(S);
    double ps = 0.0;
    for (int s = 0; s < S; ++s) { g->pi[s] = (float)avg[s]; ps += (double)g->pi[s]; }
    g->pi_sum = ps;
  }
  std::vector<float> piprob(A);
  for (int64_t i = 0; i < A; ++i) piprob[i] = g->pi[src[i]] * prob[i];
  build_ordering(A, S, dst, src, pdf, prob, piprob.data(), &g->h_fwd, nullptr, true, false);   // alpha: rows = dst
  build_ordering(A, S, src, dst, pdf, prob, piprob.data(), &g->h_bwd, nullptr, true, false);   // beta : rows = src
  build_ordering(A, P, pdf, src, dst, prob, piprob.data(), &g->h_gam, nullptr, true, false);   // gamma: rows = pdf
  // State-x layouts (chain_internal.h): peel one self-loop per state, then virtual states = distinct (dst, pdf) pairs
  // of the remaining arcs in (dst, pdf) order; a state nobody else enters gets one with pdf -1.
  {
    const char* peel_env = getenv("PK2_DEN_PEEL");
    const bool peel = !(peel_env && atoi(peel_env) == 0);
    g->loop_pdf.assign(S, -1);
    g->loop_prob.assign(S, 0.f);
    std::vector<int64_t> keep;
    keep.reserve(A);
    for (int64_t i = 0; i < A; ++i) {
      if (peel && src[i] == dst[i] && g->loop_pdf[src[i]] < 0 && prob[i] > 0.f) {
        g->loop_pdf[src[i]] = pdf[i];
        g->loop_prob[src[i]] = prob[i];
      } else {
        keep.push_back(i);
      }
    }
    const int64_t A2 = (int64_t)keep.size();
    std::vector<int32_t> src2(A2), dst2(A2), pdf2(A2);
    std::vector<float> prob2(A2), piprob2(A2);
    for (int64_t k = 0; k < A2; ++k) {
      const int64_t i = keep[k];
      src2[k] = src[i]; dst2[k] = dst[i]; pdf2[k] = pdf[i]; prob2[k] = prob[i]; piprob2[k] = piprob[i];
    }
    std::vector<int64_t> order(A2);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) {
      return dst2[x] != dst2[y] ? dst2[x] < dst2[y] : pdf2[x] < pdf2[y];
    });
    std::vector<int32_t> arc_v(A2);
    g->voff.assign(S + 1, 0);
    g->vpdf.clear();
    std::vector<int32_t> vstate;
    int64_t k = 0;
    for (int d = 0; d < S; ++d) {
      g->voff[d] = (int32_t)g->vpdf.size();
      if (k == A2 || dst2[order[k]] != d) { g->vpdf.push_back(-1); vstate.push_back(d); continue; }
      while (k < A2 && dst2[order[k]] == d) {
        const int32_t q = pdf2[order[k]];
        g->vpdf.push_back(q); vstate.push_back(d);
        while (k < A2 && dst2[order[k]] == d && pdf2[order[k]] == q) arc_v[order[k++]] = (int32_t)g->vpdf.size() - 1;
      }
    }
    g->V = (int32_t)g->vpdf.size();
    g->voff[S] = g->V;
    // occupancy states: the virtual states of d, then its peeled loop
    g->ooff.assign(S + 1, 0);
    g->opdf.clear(); g->ovirt.clear();
    for (int d = 0; d < S; ++d) {
      g->ooff[d] = (int32_t)g->opdf.size();
      for (int v = g->voff[d]; v < g->voff[d + 1]; ++v) { g->opdf.push_back(g->vpdf[v]); g->ovirt.push_back(g->voff[d]); }
      if (g->loop_pdf[d] >= 0) { g->opdf.push_back(g->loop_pdf[d]); g->ovirt.push_back(g->voff[d]); }
    }
    g->Vo = (int32_t)g->opdf.size();
    g->ooff[S] = g->Vo;
    // the 8-byte-record orderings of the state-x kernels
    build_ordering(A2, g->V, arc_v.data(), src2.data(), pdf2.data(), prob2.data(), piprob2.data(), &g->h_fwdv, vstate.data(),
                   false, true, true);
    build_ordering(A2, S, src2.data(), arc_v.data(), pdf2.data(), prob2.data(), piprob2.data(), &g->h_bwdv, nullptr, false,
                   true, true);
    build_persist(A2, g->V, arc_v.data(), src2.data(), prob2.data(), piprob2.data(), vstate.data(), S, &g->h_pfwd);
    build_persist(A2, S, src2.data(), arc_v.data(), prob2.data(), piprob2.data(), nullptr, S, &g->h_pbwd);
    build_persist2(g, A2, arc_v.data(), src2.data(), prob2.data(), piprob2.data(), vstate.data());
    if (g->h_pbwd.ok) {     // the backward workgroups also stage x for their own virtual states: max_groups = how many
      g->h_pbwd.max_groups = 0;
      for (int r = 0; r < kPR; ++r)
        g->h_pbwd.max_groups = std::max(g->h_pbwd.max_groups, g->voff[g->h_pbwd.row_begin[r + 1]] - g->voff[g->h_pbwd.row_begin[r]]);
      if (g->h_pbwd.max_groups > kPMaxRows) g->h_pbwd.ok = false;
    }
    g->po_off.assign(P + 1, 0);
    for (int o = 0; o < g->Vo; ++o) if (g->opdf[o] >= 0) g->po_off[g->opdf[o] + 1]++;
    for (int p = 0; p < P; ++p) g->po_off[p + 1] += g->po_off[p];
    g->po_occ.assign(std::max(1, g->po_off[P]), 0);
    std::vector<int32_t> cur(g->po_off.begin(), g->po_off.end() - 1);
    for (int o = 0; o < g->Vo; ++o) if (g->opdf[o] >= 0) g->po_occ[cur[g->opdf[o]]++] = o;
  }
  *out = g;
  return PK2_OK;
}

// State numbering: the persistent kernels gather from LDS and only need the rows of their 32 workgroups balanced, which the
// caller's numbering gives for any graph without a degree trend along the state ids (the in-degree order does the opposite:
// its last workgroups would own thousands of rows); the launch-per-frame kernels gather from L2 and gain from the in-degree
// order.  So: the caller's numbering when a persistent layout fits with it, else the in-degree order.
// PK2_DEN_ORDER = none | indegree | degree forces one.
static int build_graph(int32_t S, int32_t P, int64_t A, const int32_t* src_in, const int32_t* dst_in,
                       const int32_t* pdf, const float* prob, int32_t start, pk2_den_graph** out) {
  const char* env = getenv("PK2_DEN_ORDER");
  if (env) return build_graph_ordered(S, P, A, src_in, dst_in, pdf, prob, start, env, out);
  int rc = build_graph_ordered(S, P, A, src_in, dst_in, pdf, prob, start, "none", out);
  if (rc) return rc;
  if (den_persist_fits(*out) || den_persist2_fits(*out)) return PK2_OK;
  delete *out;
  *out = nullptr;
  return build_graph_ordered(S, P, A, src_in, dst_in, pdf, prob, start, "indegree", out);
}

template <typename T>
static int upload_vec(pk2_den_graph* g, const std::vector<T>& v, const T** dptr) {
  void* d = nullptr;
  PK2_HIP(hipMalloc(&d, std::max<size_t>(16, v.size() * sizeof(T))));
  g->allocs.push_back(d);
  if (!v.empty()) PK2_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *dptr = static_cast<const T*>(d);
  return PK2_OK;
}

static int upload_persist(pk2_den_graph* g, const HostPersist& h, DevPersist* d) {
  if (!h.ok) return PK2_OK;
  int rc;
  if ((rc = upload_vec(g, h.prob, &d->prob))) return rc;
  if ((rc = upload_vec(g, h.idx2, &d->idx2))) return rc;
  if ((rc = upload_vec(g, h.ends, &d->ends))) return rc;
  if ((rc = upload_vec(g, h.first_row, &d->first_row))) return rc;
  if ((rc = upload_vec(g, h.wcrow, &d->wcrow))) return rc;
  if ((rc = upload_vec(g, h.row_begin, &d->row_begin))) return rc;
  if ((rc = upload_vec(g, h.grp_begin, &d->grp_begin))) return rc;
  if ((rc = upload_vec(g, h.row_leak, &d->row_leak))) return rc;
  if ((rc = upload_vec(g, h.row_psum, &d->row_psum))) return rc;
  d->max_rows = h.max_rows; d->max_groups = h.max_groups; d->estep = h.estep;
  return PK2_OK;
}

static int upload_persist2(pk2_den_graph* g, const HostPersist2& h, DevPersist2* d) {
  if (!h.ok) return PK2_OK;
  int rc;
  if ((rc = upload_vec(g, h.prob, &d->prob))) return rc;
  if ((rc = upload_vec(g, h.idx2, &d->idx2))) return rc;
  if ((rc = upload_vec(g, h.ends, &d->ends))) return rc;
  if ((rc = upload_vec(g, h.first_row, &d->first_row))) return rc;
  if ((rc = upload_vec(g, h.uncovered, &d->uncovered))) return rc;
  if ((rc = upload_vec(g, h.ncomp, &d->ncomp))) return rc;
  if ((rc = upload_vec(g, h.rmap, &d->rmap))) return rc;
  d->num_rows = (int)(h.rmap.size() / 2);
  if ((rc = upload_vec(g, h.pbeg, &d->pbeg))) return rc;
  if ((rc = upload_vec(g, h.sprob, &d->sprob))) return rc;
  if ((rc = upload_vec(g, h.sidx2, &d->sidx2))) return rc;
  if ((rc = upload_vec(g, h.sends, &d->sends))) return rc;
  if ((rc = upload_vec(g, h.sfirst_row, &d->sfirst_row))) return rc;
  if ((rc = upload_vec(g, h.wcrow, &d->wcrow))) return rc;
  if ((rc = upload_vec(g, h.row_begin, &d->row_begin))) return rc;
  if ((rc = upload_vec(g, h.grp_begin, &d->grp_begin))) return rc;
  if ((rc = upload_vec(g, h.row_leak, &d->row_leak))) return rc;
  if ((rc = upload_vec(g, h.row_psum, &d->row_psum))) return rc;
  d->estep = h.estep; d->K = h.K; d->R = h.R;
  for (int c = 0; c <= kMaxChunks; ++c) d->cbeg[c] = h.cbeg[c];
  for (int c = 0; c < kMaxChunks; ++c) d->lds_off[c] = h.lds_off[c];
  return PK2_OK;
}

static int upload_ordering(pk2_den_graph* g, const HostOrdering& h, DevOrdering* d) {
  int rc;
  if ((rc = upload_vec(g, h.arcs, &d->arcs))) return rc;
  if ((rc = upload_vec(g, h.arcs2, &d->arcs2))) return rc;
  if ((rc = upload_vec(g, h.row_leak, &d->row_leak))) return rc;
  if ((rc = upload_vec(g, h.slot0, &d->slot0))) return rc;
  if ((rc = upload_vec(g, h.meta, &d->meta))) return rc;
  if ((rc = upload_vec(g, h.wb_off, &d->wb_off))) return rc;
  if ((rc = upload_vec(g, h.row0, &d->row0))) return rc;
  if ((rc = upload_vec(g, h.nrows, &d->nrows))) return rc;
  if ((rc = upload_vec(g, h.atomic, &d->atomic))) return rc;
  if ((rc = upload_vec(g, h.real0, &d->real0))) return rc;
  if ((rc = upload_vec(g, h.nreal, &d->nreal))) return rc;
  if ((rc = upload_vec(g, h.wb_crow, &d->wb_crow))) return rc;
  d->n_chunks = h.n_chunks;
  return PK2_OK;
}

int den_upload(pk2_den_graph* g) {
  if (g->uploaded) return PK2_OK;
  int rc;
  PK2_HIP(hipGetDevice(&g->device));
  if ((rc = upload_ordering(g, g->h_fwd, &g->fwd))) return rc;
  if ((rc = upload_ordering(g, g->h_bwd, &g->bwd))) return rc;
  if ((rc = upload_ordering(g, g->h_gam, &g->gam))) return rc;
  if ((rc = upload_ordering(g, g->h_fwdv, &g->fwdv))) return rc;
  if ((rc = upload_ordering(g, g->h_bwdv, &g->bwdv))) return rc;
  if ((rc = upload_persist(g, g->h_pfwd, &g->pfwd))) return rc;
  if ((rc = upload_persist(g, g->h_pbwd, &g->pbwd))) return rc;
  if ((rc = upload_persist2(g, g->h_p2fwd, &g->p2fwd))) return rc;
  if ((rc = upload_persist2(g, g->h_p2bwd, &g->p2bwd))) return rc;
  if ((rc = upload_vec(g, g->voff, &g->d_voff))) return rc;
  if ((rc = upload_vec(g, g->vpdf, &g->d_vpdf))) return rc;
  if ((rc = upload_vec(g, g->loop_pdf, &g->d_loop_pdf))) return rc;
  if ((rc = upload_vec(g, g->loop_prob, &g->d_loop_prob))) return rc;
  if ((rc = upload_vec(g, g->ooff, &g->d_ooff))) return rc;
  if ((rc = upload_vec(g, g->opdf, &g->d_opdf))) return rc;
  if ((rc = upload_vec(g, g->ovirt, &g->d_ovirt))) return rc;
  if ((rc = upload_vec(g, g->po_off, &g->d_po_off))) return rc;
  if ((rc = upload_vec(g, g->po_occ, &g->d_po_occ))) return rc;
  const float* dpi = nullptr;
  {   // (padded to whole 1 KB rows: the persistent kernels copy it into LDS with 16-byte LDS-DMA granules)
    std::vector<float> pi_pad(g->pi);
    pi_pad.resize((pi_pad.size() + 255) / 256 * 256 + 256, 0.f);
    if ((rc = upload_vec(g, pi_pad, &dpi))) return rc;
  }
  g->d_pi = const_cast<float*>(dpi);
  g->uploaded = true;
  return PK2_OK;
}

}  // namespace pk2

// ----------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------
using namespace pk2;

extern "C" int pk2_den_graph_create(int32_t num_states, int32_t num_pdfs, int64_t num_arcs,
                                    const int32_t* arc_src, const int32_t* arc_dst,
                                    const int32_t* arc_pdf, const float* arc_prob,
                                    int32_t start_state, pk2_den_graph** out) {
  PK2_REQUIRE(arc_src && arc_dst && arc_pdf && arc_prob && out, "den graph: null pointer");
  return build_graph(num_states, num_pdfs, num_arcs, arc_src, arc_dst, arc_pdf, arc_prob,
                     start_state, out);
}

// den.fst in OpenFst binary form (openfst_io.h): ilabel = pdf + 1, weight = -log prob.
extern "C" int pk2_den_graph_from_openfst(const char* path, int32_t num_pdfs, pk2_den_graph** out) {
  PK2_REQUIRE(path && out, "den graph: null pointer");
  FstArrays fst;
  const std::string why = read_openfst(path, &fst);
  if (!why.empty()) { set_error("%s: %s", path, why.c_str()); return PK2_ERR_IO; }
  std::vector<int32_t> pdf(fst.ilabel.size());
  std::vector<float> prob(fst.ilabel.size());
  for (size_t a = 0; a < fst.ilabel.size(); ++a) {
    if (fst.ilabel[a] <= 0) { set_error("%s: epsilon / negative ilabel in den.fst", path); return PK2_ERR_IO; }
    pdf[a] = fst.ilabel[a] - 1;
    prob[a] = expf(-fst.weight[a]);
  }
  return build_graph((int32_t)fst.num_states, num_pdfs, (int64_t)pdf.size(), fst.src.data(), fst.dst.data(),
                     pdf.data(), prob.data(), (int32_t)fst.start, out);
}

extern "C" int pk2_den_graph_destroy(pk2_den_graph* g) {
  if (!g) return PK2_OK;
  for (void* p : g->allocs) (void)hipFree(p);
  delete g;
  return PK2_OK;
}

extern "C" int pk2_den_graph_info(const pk2_den_graph* g, int32_t* num_states, int32_t* num_pdfs,
                                  int64_t* num_arcs) {
  PK2_REQUIRE(g, "den graph: null handle");
  if (num_states) *num_states = g->S;
  if (num_pdfs) *num_pdfs = g->P;
  if (num_arcs) *num_arcs = g->A;
  return PK2_OK;
}

extern "C" int pk2_den_graph_initial_probs(const pk2_den_graph* g, float* host_out) {
  PK2_REQUIRE(g && host_out, "den graph: null pointer");
  for (int32_t k = 0; k < g->S; ++k) host_out[g->orig_of[k]] = g->pi[k];      // back to the caller's state numbering
  return PK2_OK;
}

// Test hook: arcs per lane of a wave block (kK), needed to decode the orderings below.
extern "C" int32_t pk2_den_graph_arcs_per_lane(void) { return kK; }

// Test hook: copies one host-side ordering out (which: 0 = by dst, 1 = by src, 2 = by pdf; the state-x kernels' orderings:
// 3 = by virtual destination state, 4 = by src gathering virtual destination states; their 8-byte records {a, prob} come
// out as {a, 0, prob, 0}).
// Sizes are queried by passing null buffers.  meta_out receives two uint32 per lane: {first row, flush mask}.
extern "C" int pk2_den_graph_debug_ordering(const pk2_den_graph* g, int which, int64_t* n_arcs_padded,
                                            int32_t* n_chunks, int32_t* arcs_out /* int4 */,
                                            uint32_t* meta_out, int32_t* wb_off_out,
                                            int32_t* row0_out, int32_t* nrows_out,
                                            int32_t* atomic_out) {
  PK2_REQUIRE(g && which >= 0 && which < 5, "debug ordering: bad args");
  const HostOrdering* hs[5] = {&g->h_fwd, &g->h_bwd, &g->h_gam, &g->h_fwdv, &g->h_bwdv};
  const HostOrdering& h = *hs[which];
  const size_t n = which < 3 ? h.arcs.size() : h.arcs2.size();
  if (n_arcs_padded) *n_arcs_padded = (int64_t)n;
  if (n_chunks) *n_chunks = h.n_chunks;
  if (arcs_out && which < 3) memcpy(arcs_out, h.arcs.data(), n * sizeof(int4));
  if (arcs_out && which >= 3)
    for (size_t i = 0; i < n; ++i) {
      arcs_out[4 * i] = h.arcs2[i].x; arcs_out[4 * i + 1] = 0; arcs_out[4 * i + 2] = h.arcs2[i].y; arcs_out[4 * i + 3] = 0;
    }
  if (meta_out) memcpy(meta_out, h.meta.data(), h.meta.size() * sizeof(uint2));
  if (wb_off_out) memcpy(wb_off_out, h.wb_off.data(), h.wb_off.size() * sizeof(int32_t));
  if (row0_out) memcpy(row0_out, h.row0.data(), h.row0.size() * sizeof(int32_t));
  if (nrows_out) memcpy(nrows_out, h.nrows.data(), h.nrows.size() * sizeof(int32_t));
  if (atomic_out) memcpy(atomic_out, h.atomic.data(), h.atomic.size() * sizeof(int32_t));
  return PK2_OK;
}

// Test hook: the virtual states (chain_internal.h) and the per-chunk extras of ordering `which` (3 or 4): voff_out[S+1],
// vpdf_out[V], real0_out / nreal_out / slot0_out [n_chunks], row_leak_out [sum of nrows], the peeled self-loops
// loop_pdf_out[S] / loop_prob_out[S], ooff_out[S+1], opdf_out[Vo].  Null buffers are skipped; *num_virtual, *num_occ and
// *n_row_leak return the sizes.
extern "C" int pk2_den_graph_debug_virtual(const pk2_den_graph* g, int which, int32_t* num_virtual, int32_t* voff_out,
                                           int32_t* vpdf_out, int32_t* real0_out, int32_t* nreal_out,
                                           int32_t* slot0_out, int64_t* n_row_leak, float* row_leak_out,
                                           int32_t* loop_pdf_out, float* loop_prob_out, int32_t* num_occ,
                                           int32_t* ooff_out, int32_t* opdf_out) {
  PK2_REQUIRE(g && (which == 3 || which == 4), "debug virtual: bad args");
  const HostOrdering& h = which == 3 ? g->h_fwdv : g->h_bwdv;
  if (num_virtual) *num_virtual = g->V;
  if (num_occ) *num_occ = g->Vo;
  if (voff_out) memcpy(voff_out, g->voff.data(), g->voff.size() * sizeof(int32_t));
  if (vpdf_out) memcpy(vpdf_out, g->vpdf.data(), g->vpdf.size() * sizeof(int32_t));
  if (real0_out) memcpy(real0_out, h.real0.data(), h.real0.size() * sizeof(int32_t));
  if (nreal_out) memcpy(nreal_out, h.nreal.data(), h.nreal.size() * sizeof(int32_t));
  if (slot0_out) memcpy(slot0_out, h.slot0.data(), h.slot0.size() * sizeof(int32_t));
  if (n_row_leak) *n_row_leak = (int64_t)h.row_leak.size();
  if (row_leak_out) memcpy(row_leak_out, h.row_leak.data(), h.row_leak.size() * sizeof(float));
  if (loop_pdf_out) memcpy(loop_pdf_out, g->loop_pdf.data(), g->loop_pdf.size() * sizeof(int32_t));
  if (loop_prob_out) memcpy(loop_prob_out, g->loop_prob.data(), g->loop_prob.size() * sizeof(float));
  if (ooff_out) memcpy(ooff_out, g->ooff.data(), g->ooff.size() * sizeof(int32_t));
  if (opdf_out) memcpy(opdf_out, g->opdf.data(), g->opdf.size() * sizeof(int32_t));
  return PK2_OK;
}

// Test hook: the second persistent layout of an ordering (which: 0 forward, 1 backward).  info[32] = {ok, estep, K, R,
// tfloats, max_rows, max_groups, pieces, cap, rows, cbeg[0..kMaxChunks] at 10.., row arrays of the LDS layout (8: with the pdfs
// the x gather needs, 7: without) at 19, lds_off[0..kMaxChunks-1] at 20.., kPR, kPT, kPK, kPW, kSP, kSegs at 26..}; the arrays (may be null) are sized from it (chain_internal.h: HostPersist2).
extern "C" int pk2_den_graph_debug_persist2(const pk2_den_graph* g, int which, int32_t* info, float* prob, uint32_t* idx2,
                                            uint32_t* ends, int32_t* first_row, int32_t* uncovered, int32_t* ncomp, int16_t* rmap,
                                            int32_t* pbeg, float* sprob,
                                            uint32_t* sidx2, uint32_t* sends, int32_t* sfirst_row, int32_t* wcrow,
                                            int32_t* row_begin, int32_t* grp_begin, float* row_leak, float* row_psum) {
  PK2_REQUIRE(g && (which == 0 || which == 1) && info, "den graph debug: bad arguments");
  const HostPersist2& h = which == 0 ? g->h_p2fwd : g->h_p2bwd;
  for (int k = 0; k < 32; ++k) info[k] = 0;
  info[0] = h.ok ? 1 : 0; info[1] = h.estep; info[2] = h.K; info[3] = h.R; info[4] = h.tfloats; info[5] = h.max_rows;
  info[6] = h.max_groups; info[7] = (int32_t)(h.sends.size() / kPT); info[8] = g->p2_cap; info[9] = (int32_t)h.row_leak.size();
  for (int c = 0; c <= kMaxChunks; ++c) info[10 + c] = h.cbeg[c];
  static_assert(10 + kMaxChunks < 19, "info[19] is free");
  info[19] = g->p2_rowarrays;
  for (int c = 0; c < kMaxChunks; ++c) info[20 + c] = h.lds_off[c];
  info[26] = kPR; info[27] = kPT; info[28] = kPK; info[29] = kPW; info[30] = kSP; info[31] = kSegs;
  if (!h.ok) return PK2_OK;
  auto cp = [](void* dst, const void* src, size_t bytes) { if (dst && bytes) memcpy(dst, src, bytes); };
  cp(prob, h.prob.data(), h.prob.size() * 4); cp(idx2, h.idx2.data(), h.idx2.size() * 4);
  cp(ends, h.ends.data(), h.ends.size() * 4); cp(first_row, h.first_row.data(), h.first_row.size() * 4);
  cp(uncovered, h.uncovered.data(), h.uncovered.size() * 4); cp(pbeg, h.pbeg.data(), h.pbeg.size() * 4);
  cp(ncomp, h.ncomp.data(), h.ncomp.size() * 4); cp(rmap, h.rmap.data(), h.rmap.size() * 2);
  cp(sprob, h.sprob.data(), h.sprob.size() * 4); cp(sidx2, h.sidx2.data(), h.sidx2.size() * 4);
  cp(sends, h.sends.data(), h.sends.size() * 4); cp(sfirst_row, h.sfirst_row.data(), h.sfirst_row.size() * 4);
  cp(wcrow, h.wcrow.data(), h.wcrow.size() * 4); cp(row_begin, h.row_begin.data(), h.row_begin.size() * 4);
  cp(grp_begin, h.grp_begin.data(), h.grp_begin.size() * 4); cp(row_leak, h.row_leak.data(), h.row_leak.size() * 4);
  cp(row_psum, h.row_psum.data(), h.row_psum.size() * 4);
  return PK2_OK;
}

// Test hook: the persistent kernel's layout of an ordering (which: 0 = forward, rows = virtual destination states;
// 1 = backward, rows = source states).  info = {ok, max_rows, max_groups, workgroups, threads, slots per thread, waves,
// rows, estep}; the arrays (may be null) are sized from it: arcs [workgroups * slots * threads][2], ends / first_row
// [workgroups * threads], wcrow [workgroups * waves], row_begin / grp_begin [workgroups + 1], row_leak / row_psum [rows].
extern "C" int pk2_den_graph_debug_persist(const pk2_den_graph* g, int which, int32_t* info, int32_t* arcs_out,
                                           uint64_t* ends_out, int32_t* first_row_out, int32_t* wcrow_out,
                                           int32_t* row_begin_out, int32_t* grp_begin_out, float* row_leak_out,
                                           float* row_psum_out) {
  PK2_REQUIRE(g && (which == 0 || which == 1) && info, "den graph debug: bad arguments");
  const HostPersist& h = which == 0 ? g->h_pfwd : g->h_pbwd;
  info[0] = h.ok ? 1 : 0; info[1] = h.max_rows; info[2] = h.max_groups;
  info[3] = kPR; info[4] = kPT; info[5] = kPK; info[6] = kPW; info[7] = (int32_t)h.row_leak.size(); info[8] = h.estep;
  if (!h.ok) return PK2_OK;
  if (arcs_out) memcpy(arcs_out, h.arcs.data(), h.arcs.size() * sizeof(int2));
  if (ends_out) memcpy(ends_out, h.ends.data(), h.ends.size() * sizeof(uint64_t));
  if (first_row_out) memcpy(first_row_out, h.first_row.data(), h.first_row.size() * sizeof(int32_t));
  if (wcrow_out) memcpy(wcrow_out, h.wcrow.data(), h.wcrow.size() * sizeof(int32_t));
  if (row_begin_out) memcpy(row_begin_out, h.row_begin.data(), h.row_begin.size() * sizeof(int32_t));
  if (grp_begin_out) memcpy(grp_begin_out, h.grp_begin.data(), h.grp_begin.size() * sizeof(int32_t));
  if (row_leak_out) memcpy(row_leak_out, h.row_leak.data(), h.row_leak.size() * sizeof(float));
  if (row_psum_out) memcpy(row_psum_out, h.row_psum.data(), h.row_psum.size() * sizeof(float));
  return PK2_OK;
}
