// host_plan.cpp -- see host_plan.hpp.  Host-only C++17; no device code here.
#include "host_plan.hpp"

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <thread>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <pthread.h>
#include <cstring>
#include <deque>
#include <numeric>
#include <unordered_map>

namespace fmmbem {

// ---- host_parallel: a small persistent pool (host_plan.hpp) ----
namespace {
class HostPool {
 public:
  // heap-allocated and never destroyed: in a forked child the condition variables still count the parent's waiting workers, and
  // destroying one (pthread_cond_destroy) would wait for them for ever.  The workers are stopped and joined at exit instead.
  static HostPool& instance() {
    static HostPool* p = [] { HostPool* q = new HostPool; std::atexit([] { instance().shutdown(); }); return q; }();
    return *p;
  }
  void run(int nt, const std::function<void(int)>& body) {
    if (nt <= 1) { body(0); return; }
    if (inside_) { for (int t = 0; t < nt; ++t) body(t); return; }      // a fan-out inside a share runs in place
    struct Inside { Inside() { inside_ = true; } ~Inside() { inside_ = false; } } mark_inside;
    std::unique_lock<std::mutex> own(busy_, std::try_to_lock);
    if (!own.owns_lock() || dead_.load() || !start(nt)) { spawn(nt, body); return; }
    {
      std::lock_guard<std::mutex> lk(m_);
      body_ = &body; nt_ = nt; next_.store(1);
      ++epoch_;
    }
    cv_.notify_all();
    body(0);
    for (;;) {                                           // the caller takes shares too
      const int t = next_.fetch_add(1);
      if (t >= nt) break;
      body(t);
    }
    // every share has been taken; wait for the workers that are still inside one.  (A worker that wakes up after this finds no
    // job: body_ is cleared under the lock it reads it under.)
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return active_ == 0; });
    body_ = nullptr;
  }
  void shutdown() {
    if (dead_.load()) return;                            // a forked child: the threads these objects name are not in this process
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& w : workers_) if (w.joinable()) w.join();
  }

 private:
  HostPool() { pthread_atfork(nullptr, nullptr, [] { instance().dead_.store(true); }); }
  static void spawn(int nt, const std::function<void(int)>& body) {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back([&body, t] { inside_ = true; body(t); });
    body(0);
    for (auto& th : pool) th.join();
  }
  bool start(int nt) {                                   // workers on demand, at most 15
    const int want = std::min(nt - 1, 15);
    try {
      while ((int)workers_.size() < want) workers_.emplace_back([this] { work(); });
    } catch (...) { return !workers_.empty(); }
    return true;
  }
  void work() {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)>* body;
      int nt;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
        if (stop_) return;
        seen = epoch_; body = body_; nt = nt_;
        if (!body) continue;
        ++active_;
      }
      inside_ = true;
      for (;;) {
        const int t = next_.fetch_add(1);
        if (t >= nt) break;
        (*body)(t);
      }
      {
        std::lock_guard<std::mutex> lk(m_);
        --active_;
      }
      done_.notify_all();
    }
  }
  static thread_local bool inside_;
  std::mutex busy_, m_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> workers_;
  const std::function<void(int)>* body_ = nullptr;       // under m_, like nt_, epoch_, active_, stop_
  int nt_ = 0, active_ = 0;
  std::atomic<int> next_{0};
  uint64_t epoch_ = 0;
  bool stop_ = false;
  std::atomic<bool> dead_{false};
};
thread_local bool HostPool::inside_ = false;
}  // namespace
void host_parallel(int nt, const std::function<void(int)>& body) { HostPool::instance().run(nt, body); }

// ------------------------------------------------------------------------------------------
// Quadrature rules (numeric data of examples/BEM/GaussQuadrature.hpp:19-185).
// Key 7 aliases the 4-point rule (:58-59); key 17 holds 16 points (:102-115); key 79 is not offered.
// ------------------------------------------------------------------------------------------
namespace {
struct RuleBuilder {
  QuadRule& r;
  void centroid(double w) { add(1. / 3, 1. / 3, 1. / 3, w); }
  void add(double a, double b, double c, double w) {
    r.pts[r.n][0] = a; r.pts[r.n][1] = b; r.pts[r.n][2] = c; r.w[r.n] = w; ++r.n;
  }
  void rot(double a, double b, double w) { add(a, b, b, w); add(b, a, b, w); add(b, b, a, w); }
  void perm(double a, double b, double c, double w) {
    add(a, b, c, w); add(a, c, b, w); add(b, a, c, w); add(b, c, a, w); add(c, a, b, w); add(c, b, a, w);
  }
};
}  // namespace

bool quad_rule(int key, QuadRule& out) {
  out.n = 0;
  RuleBuilder b{out};
  switch (key) {
    case 1: b.centroid(1.); return true;
    case 3: b.add(.5, .5, 0., 1. / 3); b.add(0., .5, .5, 1. / 3); b.add(.5, 0., .5, 1. / 3); return true;
    case 4: case 7:
      b.centroid(-27. / 48); b.add(.6, .2, .2, 25. / 48); b.add(.2, .6, .2, 25. / 48); b.add(.2, .2, .6, 25. / 48);
      return true;
    case 13:
      b.centroid(-0.149570044467682);
      b.rot(0.479308067841920, 0.260345966079040, 0.175615257433208);
      b.rot(0.869739794195568, 0.065130102902216, 0.053347235608838);
      b.perm(0.048690315425316, 0.312865496004874, 0.638444188569810, 0.077113760890257);
      return true;
    case 17:
      b.centroid(0.144315607677787);
      b.rot(0.081414823414554, 0.459292588292723, 0.095091634267285);
      b.rot(0.658861384496480, 0.170569307751760, 0.103217370534718);
      b.rot(0.898905543365938, 0.050547228317031, 0.032458497623198);
      b.perm(0.008394777409958, 0.263112829634638, 0.728492392955404, 0.027230314174435);
      return true;
    case 19:
      b.centroid(0.097135796282799);
      b.rot(0.020634961602525, 0.489682519198738, 0.031334700227139);
      b.rot(0.125820817014127, 0.437089591492937, 0.077827541004774);
      b.rot(0.623592928761935, 0.188203535619033, 0.079647738927210);
      b.rot(0.910540973211095, 0.044729513394453, 0.025577675658698);
      b.perm(0.036838412054736, 0.221962989160766, 0.741198598784498, 0.043283539377289);
      return true;
    case 25:
      b.centroid(0.090817990382754);
      b.rot(0.028844733232685, 0.485577633383657, 0.036725957756467);
      b.rot(0.781036849029926, 0.109481575485037, 0.045321059435528);
      b.perm(0.141707219414880, 0.307939838764121, 0.550352941820999, 0.072757916845420);
      b.perm(0.025003534762686, 0.246672560639903, 0.728323904597411, 0.028327242531057);
      b.perm(0.009540815400299, 0.066803251012200, 0.923655933587500, 0.009421666963733);
      return true;
    case 79:                          // GaussQuadrature.hpp:186-272 (negative weights and one point outside the triangle included)
      b.centroid(0.033057055541624);
      b.rot(-0.001900928704400, 0.500950464352200, 0.000867019185663);
      b.rot(0.023574084130543, 0.488212957934729, 0.011660052716448);
      b.rot(0.089726636099435, 0.455136681950283, 0.022876936356421);
      b.rot(0.196007481363421, 0.401996259318289, 0.030448982673938);
      b.rot(0.488214180481157, 0.255892909759421, 0.030624891725355);
      b.rot(0.647023488009788, 0.176488255995106, 0.024368057676800);
      b.rot(0.791658289326483, 0.104170855336758, 0.015997432032024);
      b.rot(0.893862072318140, 0.053068963840930, 0.007698301815602);
      b.rot(0.916762569607942, 0.041618715196029, -0.000632060497488);
      b.rot(0.976836157186356, 0.011581921406822, 0.001751134301193);
      b.perm(0.048741583664839, 0.344855770229001, 0.606402646106160, 0.016465839189576);
      b.perm(0.006314115948605, 0.377843269594854, 0.615842614456541, 0.004839033540485);
      b.perm(0.134316520547348, 0.306635479062357, 0.559048000390295, 0.025804906534650);
      b.perm(0.013973893962392, 0.249419362774742, 0.736606743262866, 0.008471091054441);
      b.perm(0.075549132909764, 0.212775724802802, 0.711675142287434, 0.018354914106280);
      b.perm(-0.008368153208227, 0.146965436053239, 0.861402717154987, 0.000704404677908);
      b.perm(0.026686063258714, 0.137726978828923, 0.835586957912363, 0.010112684927462);
      b.perm(0.010547719294141, 0.059696109149007, 0.929756171556853, 0.003573909385950);
      return true;
    default: return false;
  }
}

// ------------------------------------------------------------------------------------------
// Mesh: octahedron subdivided (recursions-1) times, vertices pushed to the unit sphere.
// ------------------------------------------------------------------------------------------
int64_t unit_sphere(int recursions, double* vertices) {
  int64_t n = 8;
  for (int i = 1; i < recursions; ++i) n *= 4;
  if (!vertices) return n;
  using V = std::array<double, 3>;
  using T = std::array<V, 3>;
  const V corner[6] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  const int face[8][3] = {{0, 4, 2}, {2, 4, 1}, {1, 4, 3}, {3, 4, 0}, {0, 2, 5}, {2, 1, 5}, {1, 3, 5}, {3, 0, 5}};
  std::vector<T> tri(8);
  for (int f = 0; f < 8; ++f) tri[f] = {corner[face[f][0]], corner[face[f][1]], corner[face[f][2]]};
  auto mid = [](const V& a, const V& b) {
    V m = {(a[0] + b[0]) * 0.5, (a[1] + b[1]) * 0.5, (a[2] + b[2]) * 0.5};
    const double len = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
    m[0] /= len; m[1] /= len; m[2] /= len;
    return m;
  };
  for (int it = 1; it < recursions; ++it) {
    std::vector<T> next;
    next.reserve(tri.size() * 4);
    for (const T& t : tri) {
      const V a = mid(t[0], t[2]), b = mid(t[0], t[1]), c = mid(t[1], t[2]);
      next.push_back({t[0], b, a});
      next.push_back({b, t[1], c});
      next.push_back({a, b, c});
      next.push_back({a, c, t[2]});
    }
    tri.swap(next);
  }
  for (size_t i = 0; i < tri.size(); ++i)
    for (int v = 0; v < 3; ++v)
      for (int k = 0; k < 3; ++k) vertices[9 * i + 3 * v + k] = tri[i][v][k];
  return n;
}

// ------------------------------------------------------------------------------------------
// Morton helpers (x lowest).  Two coders: the reference's -- 10 bits per dimension in a 32-bit key, 10 tree levels
// (include/tree/Octree.hpp:82-92) -- and a 64-bit one with 21 bits per dimension that HostPlan::build falls back to ONLY
// when a box on level 10 still holds more than ncrit bodies (the reference cannot build such a tree at all: its shift
// 3*(levels - level - 1) wraps, Octree.hpp:649).  Cells are (extent / 2^L): the first ten levels of the deep tree are the
// boxes the 10-level coder would have made, bit for bit.
// ------------------------------------------------------------------------------------------
namespace {
inline uint32_t spread3(uint32_t x) {
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}
inline uint32_t compact3(uint32_t x) {
  x &= 0x09249249u;
  x = (x | (x >> 2)) & 0x030C30C3u;
  x = (x | (x >> 4)) & 0x0300F00Fu;
  x = (x | (x >> 8)) & 0x030000FFu;
  x = (x | (x >> 16)) & 0x000003FFu;
  return x;
}
inline uint64_t spread3(uint64_t x) {              // 21 bits -> every third bit
  x &= 0x1FFFFFull;
  x = (x | (x << 32)) & 0x1F00000000FFFFull;
  x = (x | (x << 16)) & 0x1F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}
inline uint64_t compact3(uint64_t x) {
  x &= 0x1249249249249249ull;
  x = (x | (x >> 2)) & 0x10C30C30C30C30C3ull;
  x = (x | (x >> 4)) & 0x100F00F00F00F00Full;
  x = (x | (x >> 8)) & 0x1F0000FF0000FFull;
  x = (x | (x >> 16)) & 0x1F00000000FFFFull;
  x = (x | (x >> 32)) & 0x1FFFFFull;
  return x;
}
inline int level_of_key(uint32_t key) { return (31 - __builtin_clz(key)) / 3; }
inline int level_of_key(uint64_t key) { return (63 - __builtin_clzll(key)) / 3; }

// translation vector -> class number: open addressing, a few thousand entries that stay in the cache (std::unordered_map's two
// lookups per M2L pair were most of the class numbering's time)
class ClassMap {
 public:
  ClassMap() : slots_(1 << 13, Slot{{0, 0, 0}, -1}) {}
  // the class of `key`, -1 if it has none
  int find(const IVec3& key) const {
    for (size_t h = IVec3Hash()(key) & (slots_.size() - 1);; h = (h + 1) & (slots_.size() - 1)) {
      const Slot& q = slots_[h];
      if (q.val < 0) return -1;
      if (q.key == key) return q.val;
    }
  }
  // true if `key` was new (it then has the value `val`)
  bool insert(const IVec3& key, int val) {
    if (2 * (used_ + 1) > slots_.size()) grow();
    for (size_t h = IVec3Hash()(key) & (slots_.size() - 1);; h = (h + 1) & (slots_.size() - 1)) {
      Slot& q = slots_[h];
      if (q.val < 0) { q.key = key; q.val = val; ++used_; return true; }
      if (q.key == key) return false;
    }
  }

 private:
  struct Slot { IVec3 key; int val; };
  void grow() {
    std::vector<Slot> old(slots_.size() * 2, Slot{{0, 0, 0}, -1});
    old.swap(slots_);
    used_ = 0;
    for (const Slot& q : old) if (q.val >= 0) insert(q.key, q.val);
  }
  std::vector<Slot> slots_;
  size_t used_ = 0;
};

template <class Code> struct CodedT { Code code; uint32_t idx; };

// a few host threads over [0, n) in contiguous shares: body(begin, end, share)
inline int host_threads(int64_t n, int64_t grain) {
  return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency())), n / std::max<int64_t>(1, grain)));
}
template <class F>
void par_shares(int nt, int64_t n, F&& body) {
  if (nt <= 1) { body((int64_t)0, n, 0); return; }
  host_parallel(nt, [&](int t) { body(n * t / nt, n * (t + 1) / nt, t); });
}

// Octree.hpp:617-692 on codes of L bits per dimension: BFS construction with stable 8-way bucketing per box, then the box
// geometry (:334-355 through :109-113, :243-248).  Returns false when a box on level L still holds more than ncrit bodies.
template <class Code>
bool build_tree(HostPlan& hp, unsigned L, const std::vector<double>& cen, double ext_hi[3], unsigned ncrit) {
  typedef CodedT<Code> Coded;
  const int64_t n = hp.n;
  for (int k = 0; k < 3; ++k) hp.cell[k] = (ext_hi[k] - hp.pmin[k]) / std::ldexp(1.0, (int)L);
  // ---- Morton codes (Octree.hpp:118-129) ----
  std::vector<Coded> codes(n), scratch(n);
  const bool serial = n < (1 << 15) || (std::getenv("FMMBEM_TREE_SERIAL") && std::atoi(std::getenv("FMMBEM_TREE_SERIAL")) != 0);
  par_shares(serial ? 1 : host_threads(n, 1 << 14), n, [&](int64_t i0, int64_t i1, int) {
    for (int64_t i = i0; i < i1; ++i) {
      Code s[3];
      for (int k = 0; k < 3; ++k) {
        double v = cen[3 * i + k];
        v -= hp.pmin[k];
        v /= hp.cell[k];
        s[k] = (Code)(uint32_t)v;
      }
      codes[i] = {(Code)(spread3(s[0]) | (spread3(s[1]) << 1) | (spread3(s[2]) << 2)), (uint32_t)i};
    }
  });
  const bool trace_t = std::getenv("FMMBEM_BUILD_TRACE") != nullptr;
  auto tt = std::chrono::steady_clock::now();
  auto tmark = [&](const char* w) { if (!trace_t) return; const auto now = std::chrono::steady_clock::now(); std::fprintf(stderr, "  build_tree %-20s %8.2f ms\n", w, std::chrono::duration<double, std::milli>(now - tt).count()); tt = now; };
  tmark("codes");
  std::vector<Code> key(1, (Code)1);
  hp.box_parent.assign(1, 0);
  hp.box_body_begin.assign(1, 0);
  hp.box_body_end.assign(1, (int)n);
  hp.box_child_begin.assign(1, 0);
  hp.box_child_end.assign(1, 0);
  hp.box_leaf.assign(1, 0);
  hp.box_level.assign(1, 0);
  hp.level_off.assign(1, 0);
  {
    // (room for the boxes: the appends below are the serial share of both forms)
    const size_t guess = (size_t)(n / std::max(1u, ncrit / 4)) + 64;
    key.reserve(guess);
    for (auto* v : {&hp.box_parent, &hp.box_body_begin, &hp.box_body_end, &hp.box_child_begin, &hp.box_child_end, &hp.box_level}) v->reserve(guess);
    hp.box_leaf.reserve(guess);
  }
  int deepest = 0;
  if (!serial) {
    // The same tree by a route a few threads can share.  The subdivision below is a most-significant-digit radix sort that stops
    // at the leaves, every pass stable: a box's bodies end up (a) between the bodies of the boxes before and after it in Morton
    // order and (b) in their ORIGINAL order among themselves.  So: sort all bodies by (code, index) once -- a stable
    // least-significant-digit radix sort, the passes shared by the threads -- read the boxes off the sorted codes level by level
    // (a box's children begin where the digit of the next level changes), and put every leaf's bodies back in index order.
    const int nt = host_threads(n, 1 << 14);
    {
      const int total_bits = 3 * (int)L, digit = total_bits <= 32 ? 10 : 11, B = 1 << digit;
      std::vector<int64_t> hist((size_t)nt * B);
      for (int shift = 0; shift < total_bits; shift += digit) {
        std::fill(hist.begin(), hist.end(), 0);
        par_shares(nt, n, [&](int64_t i0, int64_t i1, int t) {
          int64_t* h = hist.data() + (size_t)t * B;
          for (int64_t i = i0; i < i1; ++i) ++h[(size_t)((codes[i].code >> shift) & (Code)(B - 1))];
        });
        int64_t run = 0;                               // bucket-major, then share: what keeps the pass stable
        for (int b = 0; b < B; ++b)
          for (int t = 0; t < nt; ++t) { const int64_t c = hist[(size_t)t * B + b]; hist[(size_t)t * B + b] = run; run += c; }
        par_shares(nt, n, [&](int64_t i0, int64_t i1, int t) {
          int64_t* h = hist.data() + (size_t)t * B;
          for (int64_t i = i0; i < i1; ++i) scratch[(size_t)h[(size_t)((codes[i].code >> shift) & (Code)(B - 1))]++] = codes[i];
        });
        codes.swap(scratch);
      }
    }
    tmark("radix sort");
    // level (1 .. L) of the first octal digit in which body i differs from body i - 1; L + 1: the same finest cell
    std::vector<uint8_t> dl(n, 0);
    par_shares(nt, n, [&](int64_t i0, int64_t i1, int) {
      for (int64_t i = std::max<int64_t>(i0, 1); i < i1; ++i) {
        const Code x = codes[i].code ^ codes[i - 1].code;
        dl[i] = x == 0 ? (uint8_t)(L + 1) : (uint8_t)(L - (unsigned)((sizeof(Code) == 8 ? 63 - __builtin_clzll((uint64_t)x) : 31 - __builtin_clz((uint32_t)x)) / 3));
      }
    });
    tmark("digit levels");
    struct Kids { int n; int start[8]; uint8_t oct[8]; };
    std::vector<Kids> kids;
    for (size_t lev_first = 0, lev_end = 1; lev_first < lev_end;) {     // the boxes [lev_first, lev_end) of one level
      const int lev = hp.box_level[lev_first];
      const size_t nbx = lev_end - lev_first;
      kids.assign(nbx, Kids{});
      std::atomic<bool> too_deep{false};
      // shares of boxes with about the same number of bodies
      const int ntl = nbx < 64 ? 1 : nt;
      par_shares(ntl, (int64_t)nbx, [&](int64_t k0, int64_t k1, int) {
        for (int64_t kk = k0; kk < k1; ++kk) {
          const size_t k = lev_first + (size_t)kk;
          const int b0 = hp.box_body_begin[k], b1 = hp.box_body_end[k];
          Kids& q = kids[(size_t)kk];
          q.n = 0;
          if ((unsigned)(b1 - b0) <= ncrit) continue;
          if (lev >= (int)L) { too_deep = true; continue; }
          const unsigned shift = 3 * (L - lev - 1);
          q.start[0] = b0; q.oct[0] = (uint8_t)((codes[b0].code >> shift) & 7); q.n = 1;
          for (int i = b0 + 1; i < b1; ++i)
            if (dl[i] == lev + 1) { q.start[q.n] = i; q.oct[q.n] = (uint8_t)((codes[i].code >> shift) & 7); ++q.n; }
        }
      });
      if (too_deep) return false;
      for (size_t kk = 0; kk < nbx; ++kk) {            // the children, numbered as the queue of the serial form numbers them
        const size_t k = lev_first + kk;
        const Kids& q = kids[kk];
        if (q.n == 0) { hp.box_leaf[k] = 1; continue; }
        hp.box_child_begin[k] = (int)key.size();
        for (int c = 0; c < q.n; ++c) {
          const Code kc = (Code)((key[k] << 3) | (Code)q.oct[c]);
          const int l = level_of_key(kc);
          if (l > deepest) { deepest = l; hp.level_off.push_back((int)key.size()); }
          key.push_back(kc);
          hp.box_parent.push_back((int)k);
          hp.box_body_begin.push_back(q.start[c]);
          hp.box_body_end.push_back(c + 1 < q.n ? q.start[c + 1] : hp.box_body_end[k]);
          hp.box_child_begin.push_back(0);
          hp.box_child_end.push_back(0);
          hp.box_leaf.push_back(0);
          hp.box_level.push_back(l);
        }
        hp.box_child_end[k] = (int)key.size();
      }
      lev_first = lev_end;
      lev_end = key.size();
    }
    tmark("boxes by level");
    // a leaf's bodies in their original order (what the stable passes of the serial form leave)
    {
      const int64_t nbx = (int64_t)key.size();
      par_shares(nt, nbx, [&](int64_t k0, int64_t k1, int) {
        for (int64_t k = k0; k < k1; ++k)
          if (hp.box_leaf[k])
            std::sort(codes.begin() + hp.box_body_begin[k], codes.begin() + hp.box_body_end[k], [](const Coded& a, const Coded& b) { return a.idx < b.idx; });
      });
    }
  }
  for (size_t k = 0; serial && k < key.size(); ++k) {
    const int b0 = hp.box_body_begin[k], b1 = hp.box_body_end[k];
    if ((unsigned)(b1 - b0) <= ncrit) { hp.box_leaf[k] = 1; continue; }
    const int lev = hp.box_level[k];
    if (lev >= (int)L) return false;
    const unsigned shift = 3 * (L - lev - 1);
    int count[9] = {0};
    for (int i = b0; i < b1; ++i) ++count[((codes[i].code >> shift) & 7) + 1];
    for (int c = 0; c < 8; ++c) count[c + 1] += count[c];
    int cursor[8];
    std::copy(count, count + 8, cursor);
    for (int i = b0; i < b1; ++i) scratch[b0 + cursor[(codes[i].code >> shift) & 7]++] = codes[i];
    std::copy(scratch.begin() + b0, scratch.begin() + b1, codes.begin() + b0);
    hp.box_child_begin[k] = (int)key.size();
    for (int c = 0; c < 8; ++c) {
      if (count[c + 1] == count[c]) continue;        // empty octants get no box (:666)
      const Code kc = (Code)((key[k] << 3) | (Code)c);
      const int l = level_of_key(kc);
      if (l > deepest) { deepest = l; hp.level_off.push_back((int)key.size()); }
      key.push_back(kc);
      hp.box_parent.push_back((int)k);
      hp.box_body_begin.push_back(b0 + count[c]);
      hp.box_body_end.push_back(b0 + count[c + 1]);
      hp.box_child_begin.push_back(0);
      hp.box_child_end.push_back(0);
      hp.box_leaf.push_back(0);
      hp.box_level.push_back(l);
    }
    hp.box_child_end[k] = (int)key.size();
  }
  tmark("leaf order / serial subdivision");
  hp.nboxes = (int)key.size();
  hp.level_off.push_back(hp.nboxes);
  hp.nlevels = (int)hp.level_off.size() - 1;
  hp.perm.resize(n);
  for (int64_t i = 0; i < n; ++i) hp.perm[i] = codes[i].idx;
  hp.box_key.assign(key.begin(), key.end());
  // ---- box geometry ----
  hp.box_center.resize(3 * (size_t)hp.nboxes);
  hp.box_side.resize(hp.nboxes);
  hp.box_icoord.resize(3 * (size_t)hp.nboxes);
  const double root_side = (hp.pmin[0] + std::ldexp(1.0, (int)L) * hp.cell[0]) - hp.pmin[0];
  const Code top = (Code)1 << (3 * L);
  for (int b = 0; b < hp.nboxes; ++b) {
    Code m = key[b];
    while (!(m & top)) m <<= 3;
    const Code lower = m & ~top;
    const Code ix[3] = {compact3(lower), compact3((Code)(lower >> 1)), compact3((Code)(lower >> 2))};
    const int lev = hp.box_level[b];
    for (int k = 0; k < 3; ++k) {
      const double a = hp.pmin[k] + hp.cell[k] * double(ix[k]);
      const double w = (a + hp.cell[k]) - a;
      hp.box_center[3 * b + k] = a + w * std::ldexp(1.0, (int)L - 1 - lev);
      hp.box_icoord[3 * b + k] = 2 * (int32_t)ix[k] + (int32_t)(1 << (L - lev));   // exact, in half-cells of the finest level
    }
    hp.box_side[b] = root_side / std::ldexp(1.0, lev);
  }
  tmark("perm + geometry");
  return true;
}
}  // namespace

void alloc_panels(PanelSoA& P, int64_t n, int nq) {
  P.cx.resize(n); P.cy.resize(n); P.cz.resize(n);
  P.nx.resize(n); P.ny.resize(n); P.nz.resize(n);
  P.area.resize(n); P.bc.resize(n);
  P.quad.resize((size_t)nq * 3 * n);
  P.vert.resize((size_t)9 * n);
}

// Panel(p0, p1, p2) of kernel/LaplaceSphericalBEM.hpp:64-97: centroid, normal (p2-p0) x (p1-p0) / 2A, area, the rule's points
void fill_panel(PanelSoA& P, int64_t n, int64_t i, const double* v, const QuadRule& rule, uint8_t flag) {
  const double *p0 = v, *p1 = v + 3, *p2 = v + 6;
  P.cx[i] = (p0[0] + p1[0] + p2[0]) / 3;
  P.cy[i] = (p0[1] + p1[1] + p2[1]) / 3;
  P.cz[i] = (p0[2] + p1[2] + p2[2]) / 3;
  const double a0[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  const double a1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
  const double c[3] = {a0[1] * a1[2] - a0[2] * a1[1], -(a0[0] * a1[2] - a0[2] * a1[0]), a0[0] * a1[1] - a0[1] * a1[0]};
  const double A = 0.5 * std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  P.area[i] = A;
  P.nx[i] = c[0] / 2 / A; P.ny[i] = c[1] / 2 / A; P.nz[i] = c[2] / 2 / A;
  for (int q = 0; q < rule.n; ++q)
    for (int k = 0; k < 3; ++k)
      P.quad[((size_t)q * 3 + k) * n + i] = p0[k] * rule.pts[q][0] + p1[k] * rule.pts[q][1] + p2[k] * rule.pts[q][2];
  for (int k = 0; k < 9; ++k) P.vert[(size_t)k * n + i] = v[k];
  P.bc[i] = flag;
}

std::string HostPlan::build(const HostOptions& o, int64_t n_panels, const double* vertices, const uint8_t* bc) {
  // FMMBEM_BUILD_TRACE=1: phase times of this function on stderr (tuning aid)
  const bool trace = std::getenv("FMMBEM_BUILD_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "host_plan %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  opt = o;
  n = n_panels;
  if (n <= 0 || !vertices) return "no panels";
  if (n > (int64_t(1) << 31) - 1) return "too many panels";
  if (!quad_rule(o.quad_k, rule)) return "invalid quadrature key (valid: 1 3 4 7 13 17 19 25 79)";
  if (o.p_max < 1 || o.p_max > kPmax) return "p_max out of range";
  if (!(o.theta > 0)) return "theta must be positive";
  if (o.shard_world < 1 || o.shard_rank < 0 || o.shard_rank >= o.shard_world) return "bad shard";

  // ---- panel centroids in original order (tree is built on them, LaplaceSphericalBEM.hpp:99) ----
  std::vector<double> cen(3 * n);
  // ---- bounding cube, inflated by 1+1e-6 (Octree.hpp:67-79) ----  (minima and maxima: the same whatever the order)
  double lo[3], hi[3];
  {
    const int nt = host_threads(n, 1 << 15);
    std::vector<double> part((size_t)nt * 6);
    par_shares(nt, n, [&](int64_t i0, int64_t i1, int t) {
      double l[3], h[3];
      for (int64_t i = i0; i < i1; ++i)
        for (int k = 0; k < 3; ++k) {
          const double c = (vertices[9 * i + k] + vertices[9 * i + 3 + k] + vertices[9 * i + 6 + k]) / 3;
          cen[3 * i + k] = c;
          if (i == i0) l[k] = h[k] = c;
          else { l[k] = std::min(l[k], c); h[k] = std::max(h[k], c); }
        }
      for (int k = 0; k < 3; ++k) { part[(size_t)t * 6 + k] = l[k]; part[(size_t)t * 6 + 3 + k] = h[k]; }
    });
    for (int k = 0; k < 3; ++k) { lo[k] = part[k]; hi[k] = part[3 + k]; }
    for (int t = 1; t < nt; ++t)
      for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], part[(size_t)t * 6 + k]); hi[k] = std::max(hi[k], part[(size_t)t * 6 + 3 + k]); }
  }
  const double ext = std::max({std::fabs(hi[0] - lo[0]), std::fabs(hi[1] - lo[1]), std::fabs(hi[2] - lo[2])});
  for (int k = 0; k < 3; ++k) {
    hi[k] = std::max(hi[k], lo[k] + ext * (1 + 1e-6));
    pmin[k] = lo[k];
  }
  mark("setup+bounds");
  // The reference's coder first: 32-bit keys, 10 levels -- every tree it can build comes out bit for bit.  Deeper than that
  // (many small bodies in a large box, MultipleRedBloodCell's use case, Triangulation.hpp:260-321) only the 64-bit coder
  // resolves: 21 levels.
  tree_levels_max = 10;
  if (!build_tree<uint32_t>(*this, 10, cen, hi, o.ncrit)) {
    tree_levels_max = 21;
    if (!build_tree<uint64_t>(*this, 21, cen, hi, o.ncrit)) return "octree deeper than 21 levels (coincident panel centroids?)";
  }
  mark("tree + box geometry");
  // ---- dual tree traversal (EvalInteractionLazySparse.hpp:68-110, :239-252) ----
  auto accept = [&](int s, int t) {     // DefaultMAC, radius = side/2
    const double dx = box_center[3 * s] - box_center[3 * t], dy = box_center[3 * s + 1] - box_center[3 * t + 1],
                 dz = box_center[3 * s + 2] - box_center[3 * t + 2];
    const double rhs = (box_side[s] / 2.0 + box_side[t] / 2.0) / o.theta;
    return dx * dx + dy * dy + dz * dz > rhs * rhs;
  };
  {
    // The reference walks a FIFO (EvalInteractionLazySparse.hpp:68-110): pairs are taken in the order they were queued, so the
    // queue is a sequence of GENERATIONS -- the children of generation g, in order, are generation g + 1 -- and the lists come out
    // generation by generation, pair by pair.  A generation's pairs are independent: cut into contiguous chunks for a few
    // threads, each chunk's output (next generation, P2P pairs, M2L pairs) appended in chunk order, the lists are the serial
    // ones entry for entry (tests/test_capi_host.py, test_random_meshes.py compare them with the oracle's).
    // evaluator: 0 = EvalInteractionLazySparse; 1 = EvalLocalSparse.hpp:34-86, the same traversal with accepted
    // multipoles dropped (:120-127); 2 = EvalDiagonalSparse.hpp:33-49, every leaf with itself in box order
    typedef std::pair<int, int> Pair;
    std::vector<Pair> cur, next;
    if (opt.evaluator == 2) {
      for (int b = 0; b < nboxes; ++b)
        if (box_leaf[b]) { p2p_src.push_back(b); p2p_tgt.push_back(b); }
    } else {
      cur.emplace_back(0, 0);
    }
    struct Out { std::vector<Pair> next, p2p, lr; };
    auto walk = [&](const Pair* first, const Pair* last, Out& out) {
      for (const Pair* it = first; it != last; ++it) {
        const int s = it->first, t = it->second;
        bool split_source;
        if (box_leaf[s]) {
          if (box_leaf[t]) { out.p2p.emplace_back(s, t); continue; }
          split_source = false;
        } else if (box_leaf[t]) {
          split_source = true;
        } else {
          split_source = box_side[s] > box_side[t];        // ties split the target side (:98-108)
        }
        const int open = split_source ? s : t;
        for (int c = box_child_begin[open]; c < box_child_end[open]; ++c) {
          const int ns = split_source ? c : s, nt = split_source ? t : c;
          if (accept(ns, nt)) { if (opt.evaluator == 0) out.lr.emplace_back(ns, nt); }
          else out.next.emplace_back(ns, nt);
        }
      }
    };
    const int max_threads = (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<Out> outs;
    while (!cur.empty()) {
      const int nt = cur.size() < 8192 ? 1 : std::min<int>(max_threads, (int)(cur.size() / 4096));
      outs.assign(nt, Out{});
      if (nt == 1) walk(cur.data(), cur.data() + cur.size(), outs[0]);
      else {
        host_parallel(nt, [&](int k) { walk(cur.data() + cur.size() * k / nt, cur.data() + cur.size() * (k + 1) / nt, outs[k]); });
      }
      // the shares' finds, concatenated in share order (= the serial walk's order): sizes first, then every share copies its own
      std::vector<size_t> on(nt + 1, 0), op(nt + 1, p2p_src.size()), ol(nt + 1, lr_src.size());
      for (int k = 0; k < nt; ++k) { on[k + 1] = on[k] + outs[k].next.size(); op[k + 1] = op[k] + outs[k].p2p.size(); ol[k + 1] = ol[k] + outs[k].lr.size(); }
      next.resize(on[nt]);
      p2p_src.resize(op[nt]); p2p_tgt.resize(op[nt]);
      lr_src.resize(ol[nt]); lr_tgt.resize(ol[nt]);
      auto place = [&](int k) {
        const Out& o_ = outs[k];
        std::copy(o_.next.begin(), o_.next.end(), next.begin() + on[k]);
        for (size_t i = 0; i < o_.p2p.size(); ++i) { p2p_src[op[k] + i] = o_.p2p[i].first; p2p_tgt[op[k] + i] = o_.p2p[i].second; }
        for (size_t i = 0; i < o_.lr.size(); ++i) { lr_src[ol[k] + i] = o_.lr[i].first; lr_tgt[ol[k] + i] = o_.lr[i].second; }
      };
      if (nt == 1) place(0); else host_parallel(nt, place);
      cur.swap(next);
    }
  }

  mark("traversal");
  // ---- leaves in tree order ----
  box_leaf_index.assign(nboxes, -1);
  for (int b = 0; b < nboxes; ++b)
    if (box_leaf[b]) leaf_box.push_back(b);
  std::sort(leaf_box.begin(), leaf_box.end(), [&](int a, int b) { return box_body_begin[a] < box_body_begin[b]; });
  for (int i = 0; i < (int)leaf_box.size(); ++i) box_leaf_index[leaf_box[i]] = i;
  const int nl = nleaves();

  mark("leaves");
  // ---- near field grouped by target leaf; sources ascending (EvalP2P.hpp:87 sorts the columns) ----
  near_ptr.assign(nl + 1, 0);
  for (size_t i = 0; i < p2p_tgt.size(); ++i) ++near_ptr[box_leaf_index[p2p_tgt[i]] + 1];
  for (int i = 0; i < nl; ++i) near_ptr[i + 1] += near_ptr[i];
  near_src.resize(p2p_src.size());
  {
    std::vector<int64_t> at(near_ptr.begin(), near_ptr.end() - 1);
    for (size_t i = 0; i < p2p_src.size(); ++i) near_src[at[box_leaf_index[p2p_tgt[i]]]++] = box_leaf_index[p2p_src[i]];
  }
  near_ncols.assign(nl, 0);
  near_nnz_total = 0;
  par_shares(host_threads(nl, 4096), nl, [&](int64_t t0, int64_t t1, int) {      // every leaf sorts and counts its own columns
    for (int64_t t = t0; t < t1; ++t) {
      std::sort(near_src.begin() + near_ptr[t], near_src.begin() + near_ptr[t + 1]);   // leaf index order == body order
      int cols = 0;
      for (int64_t i = near_ptr[t]; i < near_ptr[t + 1]; ++i) {
        const int sb = leaf_box[near_src[i]];
        cols += box_body_end[sb] - box_body_begin[sb];
      }
      near_ncols[t] = cols;
    }
  });
  for (int t = 0; t < nl; ++t) {
    const int tb = leaf_box[t];
    near_nnz_total += int64_t(near_ncols[t]) * (box_body_end[tb] - box_body_begin[tb]);
  }

  mark("near lists");
  // ---- which boxes need a multipole / hold a local expansion (EvalInteractionLazySparse.hpp:173-237) ----
  // need_M: every M2L source and its whole subtree; has_L: every M2L target and its whole subtree.
  need_M.assign(nboxes, 0);
  has_L.assign(nboxes, 0);
  for (int s : lr_src) need_M[s] = 1;
  for (int t : lr_tgt) has_L[t] = 1;
  for (int b = 1; b < nboxes; ++b) {            // BFS order: parents precede children
    if (need_M[box_parent[b]]) need_M[b] = 1;
    if (has_L[box_parent[b]]) has_L[b] = 1;
  }

  // ---- the reference's lazy L2L rule (EvalInteractionLazySparse.hpp:199-237) ----
  // resolve_LR_interactions walks the M2L list in traversal order; the FIRST time a box is a target it is marked and
  // propagate_local queues parent->child for every descendant that is not marked yet.  A box that was an M2L target
  // BEFORE one of its ancestors became one is therefore never given its ancestor's expansion: the far field the
  // ancestor collected is lost for that subtree (an error that does not shrink with p).  Uniform trees (the reference's
  // spheres) have no such edge; adaptive trees do.  The plan applies every edge unless opt.reference_l2l asks for the
  // reference's set; l2l_ref_omitted counts the difference either way.
  l2l_ref_edge.assign(nboxes, 0);
  l2l_ref_omitted = 0;
  {
    std::vector<uint8_t> init(nboxes, 0);
    std::vector<int> stack;
    for (int t : lr_tgt) {
      if (init[t]) continue;
      init[t] = 1;
      stack.assign(1, t);
      while (!stack.empty()) {
        const int b = stack.back();
        stack.pop_back();
        if (box_leaf[b]) continue;
        for (int ch = box_child_begin[b]; ch < box_child_end[b]; ++ch)
          if (!init[ch]) { init[ch] = 1; l2l_ref_edge[ch] = 1; stack.push_back(ch); }
      }
    }
    for (int b = 1; b < nboxes; ++b)
      if (has_L[box_parent[b]] && !l2l_ref_edge[b]) ++l2l_ref_omitted;
  }

  mark("need/has + l2l rule");
  // ---- shard: contiguous range of target leaves ----
  std::vector<int> box_owner(nboxes, 0);                 // shard whose rows hold all of the box's bodies, -1 = spans shards
  std::vector<int64_t> shard_rb;                         // first row of every shard, then n
  {
    std::vector<int> cut;
    partition_leaves(*this, o.shard_world, cut);
    if (o.shard_world > 1) {
      std::vector<int64_t> rb(o.shard_world + 1, n);
      for (int r = 0; r < o.shard_world; ++r) rb[r] = cut[r] < nl ? box_body_begin[leaf_box[cut[r]]] : n;
      for (int b = 0; b < nboxes; ++b) {
        const int r = int(std::upper_bound(rb.begin(), rb.end(), (int64_t)box_body_begin[b]) - rb.begin()) - 1;
        box_owner[b] = (box_body_end[b] <= rb[r + 1]) ? r : -1;
      }
      shard_rb = rb;
    }
    leaf_begin = cut[o.shard_rank];
    leaf_end = cut[o.shard_rank + 1];
    row_begin = leaf_begin < nl ? box_body_begin[leaf_box[leaf_begin]] : n;
    row_end = leaf_end > leaf_begin ? box_body_end[leaf_box[leaf_end - 1]] : row_begin;
    owned_L.assign(nboxes, 0);
    for (int l = leaf_begin; l < leaf_end; ++l)
      for (int b = leaf_box[l];; b = box_parent[b]) {
        if (owned_L[b]) break;
        owned_L[b] = 1;
        if (b == 0) break;
      }
    near_nnz_owned = 0;
    for (int t = leaf_begin; t < leaf_end; ++t) {
      const int tb = leaf_box[t];
      near_nnz_owned += int64_t(near_ncols[t]) * (box_body_end[tb] - box_body_begin[tb]);
    }
  }

  mark("shard");
  has_bc[0] = has_bc[1] = false;
  if (o.panels_on_device) {
    // the flags in tree order (the panels' derived geometry is computed on the device from the caller's vertices: 0.2 GB written by
    // the host and uploaded, 60-80 ms at N = 1M, otherwise), and everything the near field needs is final: hand it over
    panels.bc.resize(n);
    for (int64_t i = 0; i < n; ++i) {
      const uint8_t flag = bc ? (bc[perm[i]] ? 1 : 0) : 0;
      panels.bc[i] = flag;
      has_bc[flag] = true;
    }
    mark("flags");
    if (o.after_near_lists) {
      const std::string err = o.after_near_lists();
      if (!err.empty()) return err;
      mark("(caller: near field to the device)");
    }
  }
  // ---- operator lists ----
  const bool su = o.shard_upward && o.shard_world > 1;
  for (int b = 0; b < nboxes; ++b) {
    if (box_leaf[b] && need_M[b] && (!su || box_owner[b] == o.shard_rank)) p2m_leaves.push_back(b);
    if (box_leaf[b] && has_L[b] && owned_L[b]) l2p_leaves.push_back(b);
  }
  m2m_level_ptr.assign(1, 0);
  m2m_ops = 0;
  for (int lev = nlevels - 1; lev >= 0; --lev) {       // deepest parents first
    for (int b = level_off[lev]; b < level_off[lev + 1]; ++b)
      if (!box_leaf[b] && need_M[b] && (!su || box_owner[b] == o.shard_rank)) {
        m2m_parents.push_back(b);
        m2m_ops += box_child_end[b] - box_child_begin[b];
      }
    m2m_level_ptr.push_back((int)m2m_parents.size());
  }
  m2m_shared_ptr.assign(1, (int)m2m_parents.size());
  xch_ptr.assign(1, 0);
  if (su) {
    for (int lev = nlevels - 1; lev >= 0; --lev) {     // parents spanning shards: every shard computes them
      for (int b = level_off[lev]; b < level_off[lev + 1]; ++b)
        if (!box_leaf[b] && need_M[b] && box_owner[b] < 0) { m2m_parents.push_back(b); m2m_ops += box_child_end[b] - box_child_begin[b]; }
      m2m_shared_ptr.push_back((int)m2m_parents.size());
    }
    for (int r = 0; r < o.shard_world; ++r) {
      for (int b = 0; b < nboxes; ++b)
        if (need_M[b] && box_owner[b] == r) xch_box.push_back(b);
      xch_ptr.push_back((int)xch_box.size());
    }
    // who READS which multipoles: shard q computes L of every box that overlaps its rows, from all of that box's sources
    const int W = o.shard_world, me = o.shard_rank;
    std::vector<std::vector<uint8_t>> need((size_t)W, std::vector<uint8_t>((size_t)nboxes, 0));
    auto shard_of_row = [&](int64_t row) { return int(std::upper_bound(shard_rb.begin(), shard_rb.end(), row) - shard_rb.begin()) - 1; };
    for (size_t i = 0; i < lr_tgt.size(); ++i) {
      const int t = lr_tgt[i], s_ = lr_src[i];
      if (box_body_end[t] <= box_body_begin[t]) continue;
      const int q0 = shard_of_row(box_body_begin[t]), q1 = shard_of_row((int64_t)box_body_end[t] - 1);
      for (int q = std::max(q0, 0); q <= q1 && q < W; ++q) need[(size_t)q][(size_t)s_] = 1;
    }
    for (int b = 0; b < nboxes; ++b)                     // the parents every shard translates read all their children
      if (!box_leaf[b] && need_M[b] && box_owner[b] < 0)
        for (int c = box_child_begin[b]; c < box_child_end[b]; ++c)
          for (int q = 0; q < W; ++q) need[(size_t)q][(size_t)c] = 1;
    xsel_send_ptr.assign(1, 0); xsel_recv_ptr.assign(1, 0);
    for (int q = 0; q < W; ++q) {
      if (q != me)
        for (int b = 0; b < nboxes; ++b)
          if (need_M[b] && box_owner[b] == me && need[(size_t)q][(size_t)b]) xsel_send_box.push_back(b);
      xsel_send_ptr.push_back((int)xsel_send_box.size());
      if (q != me)
        for (int b = 0; b < nboxes; ++b)
          if (need_M[b] && box_owner[b] == q && need[(size_t)me][(size_t)b]) xsel_recv_box.push_back(b);
      xsel_recv_ptr.push_back((int)xsel_recv_box.size());
    }
  }
  l2l_level_ptr.assign(1, 0);
  l2l_ops = 0;
  for (int lev = 1; lev < nlevels; ++lev) {            // top-down
    for (int b = level_off[lev]; b < level_off[lev + 1]; ++b)
      if (has_L[box_parent[b]] && owned_L[b] && (!o.reference_l2l || l2l_ref_edge[b])) { l2l_children.push_back(b); ++l2l_ops; }
    l2l_level_ptr.push_back((int)l2l_children.size());
  }

  mark("operator lists");
  // ---- M2L grouped by target, traversal order kept within a target; translation classes ----
  // (a stable counting sort by target, the list's chunks counted and placed by a few threads: chunk k's pairs of a target go
  // behind those of the chunks before it, which is the serial order)
  m2l_ptr.assign(nboxes + 1, 0);
  m2l_pairs_owned = 0;
  const int nt_csr = lr_tgt.size() < (1u << 16) ? 1 : host_threads((int64_t)lr_tgt.size(), 1 << 15);
  std::vector<std::vector<int>> csr_at(nt_csr, std::vector<int>((size_t)nboxes, 0));      // [chunk][box]: count, then first slot
  par_shares(nt_csr, (int64_t)lr_tgt.size(), [&](int64_t i0, int64_t i1, int k) {
    std::vector<int>& c = csr_at[k];
    for (int64_t i = i0; i < i1; ++i)
      if (owned_L[lr_tgt[i]]) ++c[lr_tgt[i]];
  });
  for (int b = 0; b < nboxes; ++b) {
    int run = m2l_ptr[b];
    for (int k = 0; k < nt_csr; ++k) { const int c = csr_at[k][b]; csr_at[k][b] = run; run += c; }
    m2l_ptr[b + 1] = run;
  }
  m2l_pairs_owned = m2l_ptr[nboxes];
  m2l_src.resize(m2l_pairs_owned);
  m2l_cls.resize(m2l_pairs_owned);
  {
    // Translation classes are numbered in the order their vectors first appear in the traversal-ordered pair list.  Three steps,
    // the two long ones by a few threads: (1) every chunk of the list collects the vectors new TO IT, in order; (2) the chunks'
    // finds are merged in chunk order -- the serial numbering; (3) every pair looks its class up.
    const size_t np = lr_tgt.size();
    const int nt = np < (1u << 16) ? 1 : (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    auto vec_of = [&](size_t i) {
      const int s = lr_src[i], t = lr_tgt[i];
      return IVec3{box_icoord[3 * t] - box_icoord[3 * s], box_icoord[3 * t + 1] - box_icoord[3 * s + 1], box_icoord[3 * t + 2] - box_icoord[3 * s + 2]};
    };
    struct Found { IVec3 key; int s, t; };
    std::vector<std::vector<Found>> found(nt);
    auto scan = [&](int k) {
      ClassMap mine;
      for (size_t i = np * k / nt; i < np * (k + 1) / nt; ++i) {
        if (!owned_L[lr_tgt[i]]) continue;
        const IVec3 key = vec_of(i);                        // exact: half-cells of the finest level (up to 2^22 on a deep tree)
        if (mine.insert(key, 0)) found[k].push_back({key, lr_src[i], lr_tgt[i]});
      }
    };
    auto run = [&](auto&& fn) {
      if (nt == 1) { fn(0); return; }
      host_parallel(nt, [&](int k) { fn(k); });
    };
    run(scan);
    ClassMap cls_of;
    for (int k = 0; k < nt; ++k)
      for (const Found& f : found[k])
        if (cls_of.insert(f.key, (int)m2l_class_rep.size() / 2)) {
          m2l_class_vec.insert(m2l_class_vec.end(), {f.key.x, f.key.y, f.key.z});
          m2l_class_rep.insert(m2l_class_rep.end(), {f.s, f.t});
        }
    std::vector<int> cls_pair(np, -1);
    run([&](int k) {
      for (size_t i = np * k / nt; i < np * (k + 1) / nt; ++i)
        if (owned_L[lr_tgt[i]]) cls_pair[i] = cls_of.find(vec_of(i));
    });
    par_shares(nt_csr, (int64_t)np, [&](int64_t i0, int64_t i1, int k) {
      std::vector<int>& at = csr_at[k];
      for (int64_t i = i0; i < i1; ++i) {
        const int t = lr_tgt[i];
        if (!owned_L[t]) continue;
        const int slot = at[t]++;
        m2l_src[slot] = lr_src[i];
        m2l_cls[slot] = cls_pair[i];
      }
    });
  }

  build_rot_items();
  mark("m2l csr + classes");
  // ---- panels in tree order, SoA (LaplaceSphericalBEM.hpp:64-97) ----
  PanelSoA& P = panels;
  const int nq = rule.n;
  if (o.panels_on_device) return {};                  // (the flags were set above, ahead of the hand-over)
  alloc_panels(P, n, nq);
  // independent per panel: cut into ranges for a few host threads (half of this function's time at N = 1M when serial)
  const int nthreads = n < (1 << 16) ? 1 : (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
  std::vector<std::array<uint8_t, 2>> seen_bc(nthreads, std::array<uint8_t, 2>{0, 0});
  auto fill_range = [&](int t, int64_t i0, int64_t i1) {
    for (int64_t i = i0; i < i1; ++i) {
      const uint8_t flag = bc ? (bc[perm[i]] ? 1 : 0) : 0;
      fill_panel(P, n, i, vertices + 9 * (size_t)perm[i], rule, flag);
      seen_bc[t][flag] = 1;
    }
  };
  if (nthreads == 1) fill_range(0, 0, n);
  else {
    host_parallel(nthreads, [&](int t) { fill_range(t, n * t / nthreads, n * (t + 1) / nthreads); });
  }
  for (int t = 0; t < nthreads; ++t) { has_bc[0] = has_bc[0] || seen_bc[t][0]; has_bc[1] = has_bc[1] || seen_bc[t][1]; }
  mark("panels SoA");
  return {};
}

// Work model for one target leaf: near entries streamed + M2L pairs landing on the leaf and (shared
// equally among their leaves) on its ancestors, weighted by a pair's cost in "entries" at p = p_max.
void partition_leaves(const HostPlan& hp, int world, std::vector<int>& cut) {
  const int nl = hp.nleaves();
  cut.assign(world + 1, nl);
  cut[0] = 0;
  if (world == 1) return;
  std::vector<double> pairs_on(hp.nboxes, 0.0);
  for (int t : hp.lr_tgt) pairs_on[t] += 1.0;
  // leaves under each box
  std::vector<int> leaves_under(hp.nboxes, 0);
  for (int b = hp.nboxes - 1; b >= 0; --b) {
    if (hp.box_leaf[b]) leaves_under[b] = 1;
    if (b) leaves_under[hp.box_parent[b]] += leaves_under[b];
  }
  const double P = hp.opt.p_max;
  const double pair_cost = 0.5 * P * P * P * (P + 1) / 8.0;   // complex MACs per pair, in units of 8 near entries
  std::vector<double> w(nl);
  double total = 0;
  for (int l = 0; l < nl; ++l) {
    const int lb = hp.leaf_box[l];
    double far = 0;
    for (int b = lb;; b = hp.box_parent[b]) {
      far += pairs_on[b] / leaves_under[b];
      if (b == 0) break;
    }
    w[l] = double(hp.near_ncols[l]) * (hp.box_body_end[lb] - hp.box_body_begin[lb]) + far * pair_cost;
    total += w[l];
  }
  double acc = 0;
  int r = 1;
  for (int l = 0; l < nl && r < world; ++l) {
    acc += w[l];
    while (r < world && acc >= total * r / world) cut[r++] = l + 1;
  }
  for (; r < world; ++r) cut[r] = nl;
}

// Work list of the rotation M2L kernel: the owned pairs in box order (a target's pairs together, traversal order), cut into
// ITEMS -- runs of whole targets that one wavefront takes 64 pairs at a time.  A target may straddle the passes of its item
// (the kernel carries its chain sums across, kernels_m2l_rot.hip), so the only idle lanes are those of an item's last pass:
// each cut goes to the target boundary, among those around the nominal item length, that leaves the fewest.  Items of up to
// kRotItemPasses passes when there are enough pairs to keep every SIMD of the chip busy several times over with them, shorter
// ones on small operators.  Two cuts of the same list (the reduction does not depend on the cut, so the two give the same bits):
//   short items (2 passes) for the orders with two or four wavefronts per SIMD, where the neighbours cover an item's first pass:
//     M2L ms at N = 1M with items of 2 / 4 / 15 passes: p = 2 0.039 / 0.041 / 0.050, p = 4 0.080 / 0.086 / 0.108, p = 8 0.290 / 0.299 / 0.318
//   long items for the orders with ONE wavefront per SIMD (p >= 9): two even rounds over the chip's 1 024 SIMDs, at most 16 passes:
//     p = 9 0.451 / 0.447 / 0.437, p = 10 0.571 (4) / 0.538 (15), p = 11 0.809 / 0.803 / 0.785, p = 12 1.042 (4) / 1.017 (15)
//     (30 passes in 1 030 items, a second round of thirty items: 0.918 at p = 10)
void HostPlan::build_rot_items() {
  constexpr int kLanes = 64, kSimds = 1024;
  constexpr int kItemsWanted = 4 * 1024;
  int kRotItemPasses = 2, kRotLongMax = 16, kRotLongRounds = 0;   // rounds: 0 = by the rule below
  if (const char* e = std::getenv("FMMBEM_ROT_LONG_ROUNDS")) kRotLongRounds = std::max(0, std::atoi(e));
  // the two knobs of the cut, for the test that ANY cut of the list gives the same bits (tests/test_gpu_parity.py)
  if (const char* e = std::getenv("FMMBEM_ROT_ITEM_PASSES")) kRotItemPasses = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("FMMBEM_ROT_LONG_MAX")) kRotLongMax = std::max(1, std::atoi(e));
  rot_src.clear(); rot_cls.clear(); rot_tgt.clear(); rot_empty.clear();
  rot_item_ptr.assign(1, 0);
  rot_item_ptr_long.assign(1, 0);
  rot_passes = rot_passes_long = 0;
  std::vector<int> tg;                                 // owned targets with sources, box order
  // The pairs of the rotation kernel are the M2L pairs of the owned targets that hold a local expansion, in box order.  Where
  // that is EVERY pair of the CSR lists (any plan that is not a shard), the lists are not copied: rot_alias, and the device
  // arrays are shared too (plan.hip) -- three arrays of 4 bytes per pair less to build and to upload (~10 M pairs at N = 1M)
  rot_alias = true;
  for (int b = 0; b < nboxes && rot_alias; ++b)
    if (m2l_ptr[b + 1] > m2l_ptr[b] && !(has_L[b] && owned_L[b])) rot_alias = false;
  int64_t n_pairs = 0;
  for (int b = 0; b < nboxes; ++b) {
    if (!(has_L[b] && owned_L[b])) continue;
    if (m2l_ptr[b + 1] == m2l_ptr[b]) { rot_empty.push_back(b); continue; }
    tg.push_back(b);
    n_pairs += m2l_ptr[b + 1] - m2l_ptr[b];
    if (!rot_alias)
      for (int i = m2l_ptr[b]; i < m2l_ptr[b + 1]; ++i) { rot_src.push_back(m2l_src[i]); rot_cls.push_back(m2l_cls[i]); rot_tgt.push_back(b); }
  }
  const int64_t want = n_pairs / ((int64_t)kLanes * kItemsWanted);
  const int nominal = kLanes * (int)std::min<int64_t>(kRotItemPasses, std::max<int64_t>(1, want));
  std::vector<int> len(tg.size());
  for (size_t i = 0; i < tg.size(); ++i) len[i] = m2l_ptr[tg[i] + 1] - m2l_ptr[tg[i]];
  rot_item_ptr.clear();
  rot_passes = cut_rot_items(len, nominal, 0, rot_item_ptr);
  const int64_t all_passes = (n_pairs + kLanes - 1) / kLanes;
  // ONE round of items over the SIMDs where items of at most kRotLongMax passes cover the list (a shard of a large operator, or a
  // small operator: every item's first pass runs without operands fetched ahead, so fewer, longer items; one rank of eight of the
  // bench workload: M2L 0.094 -> 0.085 ms, one of four 0.154 -> 0.146), two even rounds above that
  if (kRotLongRounds == 0) kRotLongRounds = all_passes <= (int64_t)kRotLongMax * kSimds ? 1 : 2;
  const int64_t slots = (int64_t)kSimds * kRotLongRounds;
  int long_passes = (int)std::max<int64_t>(nominal / kLanes, std::min<int64_t>(kRotLongMax, (all_passes + slots - 1) / slots));
  for (;;) {
    rot_item_ptr_long.clear();
    rot_passes_long = cut_rot_items(len, kLanes * long_passes, 0, rot_item_ptr_long);
    // a handful of items beyond the last even round would run alone at the end (1 030 items of 30 passes: 0.92 ms against 0.54)
    const int64_t n = (int64_t)rot_item_ptr_long.size() - 1;
    if (n <= slots || long_passes >= kRotLongMax + 8 || n < kSimds) break;
    ++long_passes;
  }
}

int64_t HostPlan::cut_rot_items(const std::vector<int>& seg_len, int nominal, int pair_base, std::vector<int>& item_ptr) {
  constexpr int kLanes = 64;
  int64_t passes = 0;
  item_ptr.push_back(pair_base);
  size_t i = 0;
  while (i < seg_len.size()) {
    // boundaries after targets i, i + 1, ...: take the one with the fewest idle lanes among the lengths in
    // (nominal - 64, nominal + 64]; the first boundary at all if a single target is longer than that
    int len = 0, best_len = 0, best_idle = kLanes;
    size_t best = i, j = i;
    while (j < seg_len.size()) {
      len += seg_len[j];
      ++j;
      if (len > nominal + kLanes && best > i) break;
      const int idle = (kLanes - len % kLanes) % kLanes;
      if (best == i || (len > nominal - kLanes && (best_len <= nominal - kLanes || idle <= best_idle))) { best = j; best_len = len; best_idle = idle; }
      if (len > nominal + kLanes) break;
    }
    item_ptr.push_back(item_ptr.back() + best_len);
    passes += (best_len + kLanes - 1) / kLanes;
    i = best;
  }
  return passes;
}

}  // namespace fmmbem
