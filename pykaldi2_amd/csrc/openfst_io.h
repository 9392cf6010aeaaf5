// Reader for OpenFst binary FSTs over the tropical semiring ("standard" arcs), the on-disk form of Kaldi's
// den.fst / HCLG.fst.  Two container types:
//   "vector": per state {f32 final, i64 narcs, arcs};
//   "const":  {states array of {f32 final, u32 pos, u32 narcs, u32 niepsilons, u32 noepsilons}, arcs array},
//             each array 16-byte aligned in the file when the header's IS_ALIGNED flag (0x4) is set.
// An arc is {i32 ilabel, i32 olabel, f32 weight, i32 nextstate}.  Embedded symbol tables are not supported
// (Kaldi writes its graphs without them).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace pk2 {

struct FstArrays {
  int64_t num_states = 0, start = 0;
  std::vector<int32_t> src, dst, ilabel, olabel;
  std::vector<float> weight, final_cost;   // final_cost[s] = +inf when s is not final
};

// Returns an empty string on success, else the reason.
inline std::string read_openfst(const char* path, FstArrays* out) {
  FILE* f = fopen(path, "rb");
  if (!f) return "cannot open file";
  struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
  int64_t pos = 0;
  auto rd = [&](void* p, size_t n) { pos += (int64_t)n; return fread(p, 1, n, f) == n; };
  int32_t magic;
  if (!rd(&magic, 4) || magic != 2125659606) return "bad magic";
  auto rdstr = [&](std::string* s) {
    int32_t n;
    if (!rd(&n, 4) || n < 0 || n > 4096) return false;
    s->resize(n);
    return n == 0 || rd(&(*s)[0], n);
  };
  std::string fst_type, arc_type;
  if (!rdstr(&fst_type) || !rdstr(&arc_type)) return "bad header";
  if (arc_type != "standard") return "arc type is not 'standard'";
  int32_t version, flags; uint64_t props; int64_t start, nstates, narcs_hdr;
  if (!rd(&version, 4) || !rd(&flags, 4) || !rd(&props, 8) || !rd(&start, 8) || !rd(&nstates, 8) || !rd(&narcs_hdr, 8))
    return "short header";
  if (flags & 3) return "embedded symbol tables are not supported";
  if (nstates <= 0 || nstates > (int64_t(1) << 31) - 2) return "bad state count";
  out->num_states = nstates;
  out->start = start;
  out->final_cost.assign(nstates, 0.f);
  struct Arc { int32_t il, ol; float w; int32_t ns; };
  if (fst_type == "vector") {
    for (int64_t s = 0; s < nstates; ++s) {
      float fin; int64_t na;
      if (!rd(&fin, 4) || !rd(&na, 8) || na < 0) return "truncated state";
      out->final_cost[s] = fin;
      for (int64_t k = 0; k < na; ++k) {
        Arc a;
        if (!rd(&a, 16)) return "truncated arc";
        out->src.push_back((int32_t)s); out->dst.push_back(a.ns);
        out->ilabel.push_back(a.il); out->olabel.push_back(a.ol); out->weight.push_back(a.w);
      }
    }
  } else if (fst_type == "const") {
    const bool aligned = (flags & 4) != 0;
    auto align = [&]() {
      if (!aligned) return true;
      char pad[16];
      const int64_t n = (16 - pos % 16) % 16;
      return n == 0 || rd(pad, (size_t)n);
    };
    struct St { float fin; uint32_t pos, narcs, nieps, noeps; };
    std::vector<St> st(nstates);
    if (!align() || !rd(st.data(), sizeof(St) * (size_t)nstates)) return "truncated state array";
    if (narcs_hdr < 0) return "bad arc count";
    std::vector<Arc> arcs((size_t)narcs_hdr);
    if (!align() || (narcs_hdr > 0 && !rd(arcs.data(), sizeof(Arc) * (size_t)narcs_hdr))) return "truncated arc array";
    for (int64_t s = 0; s < nstates; ++s) {
      out->final_cost[s] = st[s].fin;
      if ((int64_t)st[s].pos + st[s].narcs > narcs_hdr) return "state arc range out of bounds";
      for (uint32_t k = 0; k < st[s].narcs; ++k) {
        const Arc& a = arcs[st[s].pos + k];
        out->src.push_back((int32_t)s); out->dst.push_back(a.ns);
        out->ilabel.push_back(a.il); out->olabel.push_back(a.ol); out->weight.push_back(a.w);
      }
    }
  } else {
    return "FST type is neither 'vector' nor 'const'";
  }
  for (int32_t d : out->dst)
    if (d < 0 || d >= nstates) return "arc destination out of range";
  return "";
}

}  // namespace pk2
