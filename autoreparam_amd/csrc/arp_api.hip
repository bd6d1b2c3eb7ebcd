// C ABI of the engine (include/autoreparam.h): model handles, sufficient
// statistics, launcher selection.  No torch types, no callbacks.
#include <math.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include <mutex>
#include "host_common.h"

namespace arp {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

static const double kHalfLog2Pi = 0.9189385332046727;

// geometry of the last arp_vi_run of this thread (arp_vi_geometry: a measurement hook)
struct ViGeometry { int v[6]; };
static thread_local ViGeometry g_vi_geometry = {{0, 0, 0, 0, 0, 0}};
static thread_local int g_vi_attempts = 0;      // launches the calling thread's last arp_vi_run needed per chunk, at most (arp_vi_attempts)
// arp_vi_run launches whose workgroups wait for each other hold this from the launch to the end of the launch: two such
// launches from two threads of a process would share the device's workgroup slots, and a group that is only partly
// resident waits for slots the other launch's waiting groups hold
static std::mutex g_vi_launch_mutex;

// Test hook (ARP_DEBUG=1 ARP_HOST_ONLY=1, announced on stderr): model handles WITHOUT a device.  Host memory stands in
// for the device tables, so that every entry point's argument validation and host-side sizing can be driven -- and run
// under a sanitizer -- on a GPU-less machine (tests/test_api_fuzz.py).  Nothing can be computed with such a handle:
// every call that gets past validation fails with HIP's own "no device" error at its first HIP call.
static bool host_only() {
  const char* e = getenv("ARP_HOST_ONLY");
  const char* d = getenv("ARP_DEBUG");
  if (!(e && e[0] == '1' && d && d[0] == '1' && d[1] == 0)) return false;
  static bool said = false;
  if (!said) { fprintf(stderr, "libautoreparam_hip: DEBUG SWITCH ARP_HOST_ONLY=1 is in effect (handles without a device: validation only)\n"); said = true; }
  return true;
}

// integer experiment switch, honoured under ARP_DEBUG=1 only and announced on stderr
static bool debug_int(const char* name, int* out) {
  const char* e = getenv(name);
  if (!e || !e[0]) return false;
  const char* d = getenv("ARP_DEBUG");
  if (!(d && d[0] == '1' && d[1] == 0)) {
    fprintf(stderr, "libautoreparam_hip: %s=%s IGNORED (experiment switch; set ARP_DEBUG=1 to enable it)\n", name, e);
    return false;
  }
  fprintf(stderr, "libautoreparam_hip: DEBUG SWITCH %s=%s is in effect\n", name, e);
  *out = atoi(e);
  return true;
}

// pick the instantiation: requested lanes-per-chain (or a default from the chain
// count) and the smallest slice size that covers `groups`.
// `exact`: the family needs NL == ceil(groups / K), rounded up to a multiple of `unit` (only a lane's last slice may be
// padding; time_series: a lane owns whole time steps, unit = 2) -- the random-stream partition depends on it.
static const LaneOps* pick(const std::vector<LaneOps>& ops, int groups, int K_req, int C, bool exact,
                           long long fill_lanes = 131072, int unit = 1) {
  auto best_for = [&](int K) -> const LaneOps* {
    const LaneOps* best = nullptr;
    const int need = ((groups + K - 1) / K + unit - 1) / unit * unit;
    for (const auto& o : ops)
      if (o.K == K && (long long)o.NL * K >= groups && (!exact || o.NL == need) &&
          (!best || o.NL < best->NL)) best = &o;
    return best;
  };
  if (K_req > 0) return best_for(K_req);
  // default: the fewest lanes per chain that still put `fill_lanes` lanes on the device -- two waves on every
  // SIMD of the 256 CUs (256 x 4 x 2 x 64 = 131072) unless the family measured better with one
  std::vector<int> Ks;
  for (const auto& o : ops) if (std::find(Ks.begin(), Ks.end(), o.K) == Ks.end()) Ks.push_back(o.K);
  std::sort(Ks.begin(), Ks.end());
  const LaneOps* last = nullptr;
  for (int K : Ks) {
    const LaneOps* o = best_for(K);
    if (!o) continue;
    last = o;
    if ((long long)C * K >= fill_lanes) return o;
  }
  return last;
}

static const std::vector<LaneOps>* family(const arp_model* m) {
  switch (m->model) {
    case ARP_MODEL_RADON: return &radon_ops();
    case ARP_MODEL_EIGHT_SCHOOLS: return &schools_ops();
    case ARP_MODEL_ELECTION: return &election_ops();
    case ARP_MODEL_GERMAN_CREDIT: return &german_ops();
    case ARP_MODEL_RADON_STDDVS: return &radon_sd_ops();
    case ARP_MODEL_NEALS_FUNNEL: return &funnel_ops();
    case ARP_MODEL_ELECTRIC: return &electric_ops();
    case ARP_MODEL_TIME_SERIES: return &time_series_ops();
    default: return nullptr;
  }
}
static const void* family_args(const arp_model* m) {
  switch (m->model) {
    case ARP_MODEL_RADON: return &m->radon;
    case ARP_MODEL_EIGHT_SCHOOLS: return &m->schools;
    case ARP_MODEL_ELECTION: return &m->election;
    case ARP_MODEL_GERMAN_CREDIT: return &m->german;
    case ARP_MODEL_RADON_STDDVS: return &m->radon_sd;
    case ARP_MODEL_NEALS_FUNNEL: return &m->funnel;
    case ARP_MODEL_ELECTRIC: return &m->electric;
    case ARP_MODEL_TIME_SERIES: return &m->time_series;
    default: return nullptr;
  }
}

static int upload_tables(arp_model* m);

static int build_radon(arp_model* m, const arp_dataset* d) {
  const int J = d->n_groups, N = d->n_obs;
  if (!d->group_host || !d->u_host || !d->x_host || !d->y_host || J <= 0 || N <= 0) {
    set_error("radon: group/u/x/y and n_groups/n_obs are required");
    return 1;
  }
  std::vector<double> n(J, 0.0), sx(J, 0.0), sy(J, 0.0);
  double sxy = 0, sxx = 0, syy = 0;
  for (int i = 0; i < N; ++i) {
    int j = d->group_host[i];
    double x = d->x_host[i], y = d->y_host[i];
    sxy += x * y; sxx += x * x; syy += y * y;
    // tf.one_hot gives an all-zero row for an out-of-range county: such an
    // observation sees no county effect and only informs b2 (models.py:834-836)
    if (j < 0 || j >= J) continue;
    n[j] += 1; sx[j] += x; sy[j] += y;
  }
  m->D = 3 + J;
  m->n_groups = J;
  m->host_tables.resize(4 * (size_t)J);
  for (int j = 0; j < J; ++j) {
    m->host_tables[j] = (float)n[j];
    m->host_tables[J + j] = (float)sx[j];
    m->host_tables[2 * J + j] = (float)sy[j];
    m->host_tables[3 * J + j] = d->u_host[j];
  }
  if (upload_tables(m)) return 1;
  m->radon.n = m->dev_tables;
  m->radon.sx = m->dev_tables + J;
  m->radon.sy = m->dev_tables + 2 * J;
  m->radon.u = m->dev_tables + 3 * J;
  m->radon.sxy = (float)sxy;
  m->radon.sxx = (float)sxx;
  double sy_tot = 0, suy_tot = 0;
  for (int j = 0; j < J; ++j) { sy_tot += sy[j]; suy_tot += (double)d->u_host[j] * sy[j]; }
  m->radon.sy_tot = (float)sy_tot;
  m->radon.suy_tot = (float)suy_tot;
  m->radon.J = J;
  // every Normal has unit scale under every (a,b): const = -(3+J+N) 0.5 log 2pi - 0.5 Syy
  m->const_base = -(3.0 + J + N) * kHalfLog2Pi - 0.5 * syy;
  return 0;
}

static int upload_tables(arp_model* m) {
  if (m->host_only) {
    m->dev_tables = (float*)malloc(std::max<size_t>(1, m->host_tables.size()) * sizeof(float));
    if (!m->dev_tables) { set_error("upload_tables: out of memory"); return 1; }
    memcpy(m->dev_tables, m->host_tables.data(), m->host_tables.size() * sizeof(float));
    return 0;
  }
  ARP_HIP_OK(hipMalloc(&m->dev_tables, m->host_tables.size() * sizeof(float)));
  ARP_HIP_OK(hipMemcpy(m->dev_tables, m->host_tables.data(), m->host_tables.size() * sizeof(float),
                       hipMemcpyHostToDevice));
  return 0;
}

// reference models.py:131-147: y_host = effects, u_host = stddevs
static int build_schools(arp_model* m, const arp_dataset* d) {
  if (!d->y_host || !d->u_host) { set_error("8schools: y (effects) and u (stddevs) are required"); return 1; }
  m->D = 10; m->n_groups = 8;
  m->host_tables.assign(d->y_host, d->y_host + 8);
  m->host_tables.insert(m->host_tables.end(), d->u_host, d->u_host + 8);
  if (upload_tables(m)) return 1;
  m->schools.y = m->dev_tables;
  m->schools.sigma = m->dev_tables + 8;
  double c = -18.0 * kHalfLog2Pi;
  for (int k = 0; k < 8; ++k) c -= log((double)d->u_host[k]);
  m->const_base = c;
  m->top_scale = {{0, log(5.0)}, {1, log(5.0)}};
  return 0;
}

// reference models.py:967-989: group = 1-based state fed to tf.one_hot(., S); x = female, x2 = black
static int build_election(arp_model* m, const arp_dataset* d) {
  const int S = d->n_groups, N = d->n_obs;
  if (!d->group_host || !d->x_host || !d->x2_host || !d->y_host || S <= 0 || N <= 0) {
    set_error("election: group/x(female)/x2(black)/y and n_groups/n_obs are required");
    return 1;
  }
  m->D = S + 4; m->n_groups = S + 1;
  std::vector<double> cn((size_t)(S + 1) * 4, 0.0), cy((size_t)(S + 1) * 4, 0.0);
  for (int i = 0; i < N; ++i) {
    int t = d->group_host[i];
    if (t < 0 || t >= S) t = S;  // all-zero one-hot row: no state effect
    int c = (d->x_host[i] != 0.0f ? 1 : 0) + (d->x2_host[i] != 0.0f ? 2 : 0);
    if ((d->x_host[i] != 0.0f && d->x_host[i] != 1.0f) || (d->x2_host[i] != 0.0f && d->x2_host[i] != 1.0f)) {
      set_error("election: female/black must be 0/1 indicators for the cell collapse");
      return 1;
    }
    cn[(size_t)t * 4 + c] += 1.0;
    cy[(size_t)t * 4 + c] += d->y_host[i];
  }
  m->host_tables.resize(cn.size() * 2);
  for (size_t i = 0; i < cn.size(); ++i) { m->host_tables[i] = (float)cn[i]; m->host_tables[cn.size() + i] = (float)cy[i]; }
  if (upload_tables(m)) return 1;
  m->election.cell_n = m->dev_tables;
  m->election.cell_y = m->dev_tables + cn.size();
  m->election.S = S;
  m->const_base = -(4.0 + S) * kHalfLog2Pi;
  m->top_scale = {{0, log(100.0)}, {1, log(10.0)}, {2 + S, log(100.0)}, {3 + S, log(100.0)}};
  return 0;
}

// reference models.py:763-806 (radon_stddvs): same inputs as radon; every county keeps all six
// second-order sufficient statistics because its observation scale is a latent
static int build_radon_sd(arp_model* m, const arp_dataset* d) {
  const int J = d->n_groups, N = d->n_obs;
  if (!d->group_host || !d->u_host || !d->x_host || !d->y_host || J <= 0 || N <= 0) {
    set_error("radon_stddvs: group/u/x/y and n_groups/n_obs are required");
    return 1;
  }
  std::vector<double> st(6 * (size_t)J, 0.0);
  for (int i = 0; i < N; ++i) {
    int j = d->group_host[i];
    if (j < 0 || j >= J) {   // a zero one-hot row would give the observation a zero scale (models.py:785-786)
      set_error("radon_stddvs: county index out of range");
      return 1;
    }
    double x = d->x_host[i], y = d->y_host[i];
    st[j] += 1; st[J + j] += x; st[2 * J + j] += y; st[3 * J + j] += x * x; st[4 * J + j] += x * y; st[5 * J + j] += y * y;
  }
  m->D = 3 + 2 * J; m->n_groups = J;
  m->host_tables.resize(7 * (size_t)J);
  for (size_t k = 0; k < 6 * (size_t)J; ++k) m->host_tables[k] = (float)st[k];
  for (int j = 0; j < J; ++j) m->host_tables[6 * (size_t)J + j] = d->u_host[j];
  if (upload_tables(m)) return 1;
  float* t = m->dev_tables;
  m->radon_sd.n = t; m->radon_sd.sx = t + J; m->radon_sd.sy = t + 2 * J; m->radon_sd.sxx = t + 3 * J;
  m->radon_sd.sxy = t + 4 * J; m->radon_sd.syy = t + 5 * J; m->radon_sd.u = t + 6 * J;
  m->radon_sd.J = J;
  m->const_base = -(3.0 + 2.0 * J + N) * kHalfLog2Pi;
  return 0;
}

// reference models.py:860-904: X = [N][F] design matrix (intercept, standardised
// numerics, one-hot blocks), y = 0/1 outcomes
static int build_german(arp_model* m, const arp_dataset* d) {
  const int N = d->n_obs, F = d->n_features;
  if (!d->X_host || !d->y_host || N <= 0 || F <= 0 || F > kGermanCols) {
    set_error("german_credit: X, y, n_obs and 0 < n_features <= 64 are required");
    return 1;
  }
  m->D = 1 + 2 * F; m->n_groups = F;
  // [N][64] rows + outcomes (the 8- and 16-lane likelihoods), then the image the matrix-core likelihood copies
  // into LDS with LDS-DMA (model_german.h, "tile image"): per 128 observations the rows with their 16-byte chunks
  // XOR-permuted, and one piece of outcomes
  const size_t plain = ((size_t)N * kGermanCols + N + 255) & ~(size_t)255;
  const int nt = (N + kGermanTileRows - 1) / kGermanTileRows;
  m->host_tables.assign(plain + (size_t)nt * kGermanImgTile, 0.0f);
  for (int n = 0; n < N; ++n)
    for (int f = 0; f < F; ++f) m->host_tables[(size_t)n * kGermanCols + f] = d->X_host[(size_t)n * F + f];
  for (int n = 0; n < N; ++n) m->host_tables[(size_t)N * kGermanCols + n] = d->y_host[n];
  for (int t = 0; t < nt; ++t) {
    float* img = m->host_tables.data() + plain + (size_t)t * kGermanImgTile;
    for (int r = 0; r < kGermanTileRows; ++r) {
      const int n = t * kGermanTileRows + r;
      if (n >= N) break;
      float* rowp = img + r * kGermanCols;   // chunk c of row r at chunk position c ^ (r & 11)
      for (int f = 0; f < F; ++f) rowp[((((f >> 2) ^ (r & 11)) & 15) << 2) + (f & 3)] = d->X_host[(size_t)n * F + f];
      img[32 * 256 + r] = d->y_host[n];
    }
  }
  // the bf16 x 3 image (model_german.h): usable when at most 8 columns are not exact in ONE bf16 piece
  const size_t f32_floats = m->host_tables.size();
  std::vector<int> split;
  for (int f = 0; f < F; ++f) {
    bool exact = true;
    for (int n = 0; n < N && exact; ++n) {
      uint32_t h, mm_, l;
      bf3_split(d->X_host[(size_t)n * F + f], h, mm_, l);
      exact = mm_ == 0u && l == 0u;
    }
    if (!exact) split.push_back(f);
  }
  const bool bf3 = (int)split.size() <= kBf3MaxSplit;
  const int ntb = (N + kBf3Rows - 1) / kBf3Rows;
  for (int q = 0; q < kBf3MaxSplit; ++q) m->german.sidx[q] = bf3 && q < (int)split.size() ? split[q] : -1;
  if (bf3) {
    m->host_tables.resize(f32_floats + (size_t)ntb * kBf3ImgTile, 0.0f);
    auto put = [](unsigned char* img, size_t byte_off, uint32_t bits) {      // the bf16 = high half of the f32 pattern
      img[byte_off] = (unsigned char)(bits >> 16); img[byte_off + 1] = (unsigned char)(bits >> 24);
    };
    for (int t = 0; t < ntb; ++t) {
      unsigned char* img = reinterpret_cast<unsigned char*>(m->host_tables.data() + f32_floats + (size_t)t * kBf3ImgTile);
      for (int r = 0; r < kBf3Rows; ++r) {
        const int n = t * kBf3Rows + r;
        if (n >= N) break;
        // where observation r sits in a backward fragment: k-step s, lane group g, element jj
        const int s_ = r >> 5, rr = r & 31;
        const int g = rr < 16 ? rr >> 2 : (rr - 16) >> 2, jj = rr < 16 ? rr & 3 : 4 + ((rr - 16) & 3);
        for (int f = 0; f < F; ++f) {
          uint32_t h, mm_, l;
          bf3_split(d->X_host[(size_t)n * F + f], h, mm_, l);
          put(img, kBf3XhF + (size_t)r * 128 + ((((size_t)f >> 3) ^ (((size_t)r >> 1) & 7)) << 4) + (f & 7) * 2, h);
          put(img, kBf3XhB + (size_t)s_ * 4096 + (size_t)f * 64 + (((size_t)g ^ (((size_t)f >> 2) & 3)) << 4) + jj * 2, h);
        }
        for (int q = 0; q < (int)split.size(); ++q) {
          uint32_t h, mm_, l;
          bf3_split(d->X_host[(size_t)n * F + split[q]], h, mm_, l);
          const size_t rowb = kBf3XaF + (size_t)r * 64, sw = ((size_t)r >> 2) & 3;
          put(img, rowb + ((0 ^ sw) << 4) + q * 2, mm_);       // [xm | xm | xl | 0]
          put(img, rowb + ((1 ^ sw) << 4) + q * 2, mm_);
          put(img, rowb + ((2 ^ sw) << 4) + q * 2, l);
          for (int part = 0; part < 2; ++part) {                // output rows q (xm) and 8 + q (xl)
            const size_t o = (size_t)part * 8 + q;
            put(img, kBf3XaB + (size_t)s_ * 1024 + o * 64 + (((size_t)g ^ ((o >> 2) & 3)) << 4) + jj * 2, part ? l : mm_);
          }
        }
        reinterpret_cast<float*>(img + kBf3Y)[r] = d->y_host[n];
      }
    }
  }
  if (upload_tables(m)) return 1;
  m->german.X = m->dev_tables;
  m->german.y = m->dev_tables + (size_t)N * kGermanCols;
  m->german.Xt = m->dev_tables + plain;
  m->german.Xb = bf3 ? m->dev_tables + f32_floats : nullptr;
  m->german.N = N; m->german.F = F;
  m->const_base = -(1.0 + 2.0 * F) * kHalfLog2Pi;
  m->top_scale = {{0, log(10.0)}};
  return 0;
}

// reference models.py:1011-1046: group = pair, group2 = grade, group3 = grade_pair (all 1-based, fed to
// tf.one_hot as they are), x = treatment, y = scores.  Observations collapse to (pair, treatment) cells.
static int build_electric(arp_model* m, const arp_dataset* d) {
  const int P = d->n_groups, N = d->n_obs, G = d->n_features;
  if (!d->group_host || !d->group2_host || !d->group3_host || !d->x_host || !d->y_host || P <= 0 || N <= 0) {
    set_error("electric: group(pair)/group2(grade)/group3(grade_pair)/x(treatment)/y and n_groups/n_obs are required");
    return 1;
  }
  if (G != kElG) { set_error("electric: n_features (n_grade = n_grade_pair) must be 4"); return 1; }
  const int R = P + 1;   // group P: observations whose pair index falls on the all-zero one-hot row
  std::vector<double> n(2 * (size_t)R, 0.0), sy(2 * (size_t)R, 0.0), syy(2 * (size_t)R, 0.0);
  std::vector<int> grade(R, -1);
  for (int i = 0; i < N; ++i) {
    int j = d->group_host[i];
    if (j < 0 || j >= P) j = P;
    int g = d->group2_host[i];
    if (g < 0 || g >= G) g = G;   // zero row: b = 0, scale exp(0)
    const float t = d->x_host[i];
    if (t != 0.0f && t != 1.0f) { set_error("electric: treatment must be a 0/1 indicator for the cell collapse"); return 1; }
    if (grade[j] >= 0 && grade[j] != g) {
      set_error("electric: the observations of one pair must share a grade for the cell collapse");
      return 1;
    }
    grade[j] = g;
    const size_t c = (size_t)(t != 0.0f) * R + j;
    const double y = d->y_host[i];
    n[c] += 1; sy[c] += y; syy[c] += y * y;
  }
  m->D = 3 * G + P; m->n_groups = R;
  m->host_tables.assign(13 * (size_t)R, 0.0f);
  float* T = m->host_tables.data();
  for (int j = 0; j < R; ++j) {
    if (j < P) {
      const int k = d->group3_host[j];
      if (k >= 0 && k < G) T[(size_t)k * R + j] = 100.0f;
    }
    if (grade[j] >= 0 && grade[j] < G) T[(size_t)(4 + grade[j]) * R + j] = 1.0f;
    double ss = 0;
    for (int t = 0; t < 2; ++t) {
      const size_t c = (size_t)t * R + j;
      const double mean = n[c] > 0 ? sy[c] / n[c] : 0.0;
      T[(size_t)(8 + 2 * t) * R + j] = (float)n[c];
      T[(size_t)(9 + 2 * t) * R + j] = (float)mean;
      ss += syy[c] - n[c] * mean * mean;
    }
    T[(size_t)12 * R + j] = (float)(ss > 0 ? ss : 0.0);
  }
  if (upload_tables(m)) return 1;
  float* t = m->dev_tables;
  m->electric.wm = t; m->electric.og = t + 4 * (size_t)R;
  m->electric.n0 = t + 8 * (size_t)R; m->electric.y0 = t + 9 * (size_t)R;
  m->electric.n1 = t + 10 * (size_t)R; m->electric.y1 = t + 11 * (size_t)R;
  m->electric.ss = t + 12 * (size_t)R;
  m->electric.P = P;
  m->const_base = -(double)(m->D + N) * kHalfLog2Pi;
  for (int k = 0; k < G; ++k) m->top_scale.push_back({2 * G + P + k, log(100.0)});
  return 0;
}

// reference models.py:1069-1141: x = regressor (years), y = series, n_obs = T
static int build_time_series(arp_model* m, const arp_dataset* d) {
  const int T = d->n_obs;
  if (!d->x_host || !d->y_host || T <= 0) { set_error("time_series: x, y and n_obs are required"); return 1; }
  // the block scan splits the T steps over the lanes of a chain in blocks of ceil(T / K) (the last lanes padded); the
  // instantiations of inst_time_series.hip are those of the reference's T = 60
  if (T != kTsSteps) { set_error("time_series: n_obs must be 60 (add TimeSeriesLane<K, 2 ceil(T / K)> to inst_time_series.hip for another length)"); return 1; }
  m->D = 3 + 2 * T; m->n_groups = 2 * T;
  m->host_tables.assign(d->x_host, d->x_host + T);
  m->host_tables.insert(m->host_tables.end(), d->y_host, d->y_host + T);
  if (upload_tables(m)) return 1;
  m->time_series.x = m->dev_tables;
  m->time_series.y = m->dev_tables + T;
  m->time_series.T = T;
  m->const_base = -(double)(m->D + T) * kHalfLog2Pi - T * log(0.12);
  return 0;
}

namespace {
// test hook (arp_adapt_probe): adapt_update on scripted log acceptance ratios, one row per thread
__global__ void adapt_probe_kernel(HmcParams P, const float* __restrict__ la, int n, float* __restrict__ adapt,
                                   float* __restrict__ kappa_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float kappa = adapt[i * 4 + 0], esum = adapt[i * 4 + 1], logavg = adapt[i * 4 + 2];
  for (int s = 0; s < P.n_steps; ++s) {
    adapt_update(P, P.step_base + s + 1, la[(size_t)s * n + i], kappa, esum, logavg);
    if (kappa_out) kappa_out[(size_t)s * n + i] = kappa;
  }
  adapt[i * 4 + 0] = kappa; adapt[i * 4 + 1] = esum; adapt[i * 4 + 2] = logavg;
}

}  // namespace

}  // namespace arp

using namespace arp;

extern "C" {

int arp_version(void) { return ARP_ABI_VERSION; }
const char* arp_last_error(void) { return g_err.c_str(); }

int arp_model_create(const arp_dataset* data, arp_model** out) {
  if (!data || !out) { set_error("arp_model_create: null argument"); return 1; }
  std::unique_ptr<arp_model> m(new arp_model());
  m->model = data->model;
  m->host_only = host_only();
  if (m->host_only) m->device = -1;
  else {
    ARP_HIP_OK(hipGetDevice(&m->device));
    hipDeviceProp_t prop;
    ARP_HIP_OK(hipGetDeviceProperties(&prop, m->device));
    m->cus = prop.multiProcessorCount;
    int coop = 0;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, m->device) == hipSuccess) m->coop_ok = coop != 0;
    ARP_HIP_OK(hipHostMalloc((void**)&m->relay_err, sizeof(unsigned), hipHostMallocMapped));
    *m->relay_err = 0u;
    const hipError_t mapped = hipHostGetDevicePointer((void**)&m->relay_err_dev, m->relay_err, 0);
    if (mapped != hipSuccess) {          // (the handle is not complete yet: nothing else to release)
      (void)hipHostFree(m->relay_err);
      m->relay_err = nullptr;
      ARP_HIP_OK(mapped);
    }
  }
  int rc;
  switch (data->model) {
    case ARP_MODEL_RADON: rc = build_radon(m.get(), data); break;
    case ARP_MODEL_EIGHT_SCHOOLS: rc = build_schools(m.get(), data); break;
    case ARP_MODEL_ELECTION: rc = build_election(m.get(), data); break;
    case ARP_MODEL_GERMAN_CREDIT: rc = build_german(m.get(), data); break;
    case ARP_MODEL_RADON_STDDVS: rc = build_radon_sd(m.get(), data); break;
    case ARP_MODEL_ELECTRIC: rc = build_electric(m.get(), data); break;
    case ARP_MODEL_TIME_SERIES: rc = build_time_series(m.get(), data); break;
    case ARP_MODEL_NEALS_FUNNEL:   // models.py:671-696: no data
      m->D = 2; m->n_groups = 1; m->const_base = -2.0 * kHalfLog2Pi; m->top_scale = {{0, log(3.0)}};
      rc = 0; break;
    default: set_error("arp_model_create: unknown model id"); return 1;
  }
  if (rc) { arp_model_destroy(m.release()); return rc; }
  for (int w = 0; w < 2; ++w) {
    if (m->host_only) {
      m->dev_ab[w] = (float*)malloc(2 * (size_t)m->D * sizeof(float));
      if (!m->dev_ab[w]) { set_error("arp_model_create: out of memory"); arp_model_destroy(m.release()); return 1; }
      continue;
    }
    if (hipMalloc(&m->dev_ab[w], 2 * (size_t)m->D * sizeof(float)) != hipSuccess) {
      set_error("arp_model_create: hipMalloc of the parameterisation arrays failed");
      arp_model_destroy(m.release());
      return 1;
    }
  }
  // default parameterisations: 0 = CP (a=b=1), 1 = NCP (a=b=0)
  std::vector<float> ones(m->D, 1.0f), zeros(m->D, 0.0f);
  // the handle is handed over only once it is complete: on any failure the caller gets *out == NULL and nothing to destroy
  *out = nullptr;
  if (arp_model_set_param(m.get(), 0, ones.data(), ones.data()) ||
      arp_model_set_param(m.get(), 1, zeros.data(), zeros.data())) {
    arp_model_destroy(m.release());
    return 1;
  }
  *out = m.release();
  return 0;
}

int arp_model_destroy(arp_model* m) {
  if (!m) return 0;
  if (m->host_only) {
    free(m->dev_tables);
    for (int w = 0; w < 2; ++w) free(m->dev_ab[w]);
    delete m;
    return 0;
  }
  if (m->dev_tables) (void)hipFree(m->dev_tables);
  for (int w = 0; w < 2; ++w) if (m->dev_ab[w]) (void)hipFree(m->dev_ab[w]);
  if (m->vi_ws) (void)hipFree(m->vi_ws);
  if (m->vi_snap) (void)hipFree(m->vi_snap);
  if (m->relay_err) (void)hipHostFree(m->relay_err);
  delete m;
  return 0;
}

int arp_model_dim(const arp_model* m) { return m ? m->D : -1; }

int arp_model_set_option(arp_model* m, const char* key, const char* value) {
  if (!m || !key || !value) { set_error("arp_model_set_option: null argument"); return 1; }
  if (!strcmp(key, "german_math")) {
    if (m->model != ARP_MODEL_GERMAN_CREDIT) { set_error("arp_model_set_option: german_math applies to german credit only"); return 1; }
    if (!strcmp(value, "auto")) m->german_math = 0;
    else if (!strcmp(value, "f32")) m->german_math = 1;
    else if (!strcmp(value, "bf16x3")) {
      if (!m->german.Xb) { set_error("arp_model_set_option: this design matrix has more than 8 columns that need three bf16 pieces"); return 1; }
      m->german_math = 2;
    } else { set_error("arp_model_set_option: german_math is one of auto, f32, bf16x3"); return 1; }
    return 0;
  }
  if (!strcmp(key, "vi_launch")) {
    // how arp_vi_run starts a kernel whose workgroups wait for each other: "cooperative" (hipLaunchCooperativeKernel),
    // "plain" (ordinary launch, one at a time per process) or "auto" (cooperative where the device supports it)
    if (!strcmp(value, "auto")) m->vi_launch = 0;
    else if (!strcmp(value, "plain")) m->vi_launch = 1;
    else if (!strcmp(value, "cooperative")) m->vi_launch = 2;
    else { set_error("arp_model_set_option: vi_launch is one of auto, plain, cooperative"); return 1; }
    return 0;
  }
  set_error("arp_model_set_option: unknown key");
  return 1;
}

double arp_model_logp_const(const arp_model* m, int which) {
  return (m && which >= 0 && which < 2) ? m->logp_const[which] : NAN;
}

int arp_model_set_param(arp_model* m, int which, const float* a_host, const float* b_host) {
  if (!m || which < 0 || which > 1 || !a_host || !b_host) { set_error("arp_model_set_param: bad argument"); return 1; }
  if (m->host_only) {
    memcpy(m->dev_ab[which], a_host, m->D * sizeof(float));
    memcpy(m->dev_ab[which] + m->D, b_host, m->D * sizeof(float));
  } else {
    ARP_HIP_OK(hipMemcpy(m->dev_ab[which], a_host, m->D * sizeof(float), hipMemcpyHostToDevice));
    ARP_HIP_OK(hipMemcpy(m->dev_ab[which] + m->D, b_host, m->D * sizeof(float), hipMemcpyHostToDevice));
  }
  m->has_param[which] = true;
  bool all1 = true, all0 = true, b1 = true;
  for (int d = 0; d < m->D; ++d) {
    all1 = all1 && a_host[d] == 1.0f && b_host[d] == 1.0f;
    all0 = all0 && a_host[d] == 0.0f && b_host[d] == 0.0f;
    b1 = b1 && b_host[d] == 1.0f;
  }
  m->param_kind[which] = all1 ? kModeCP : (all0 ? kModeNCP : (b1 ? kModeB1 : kModeVIP));
  // dropped constant: -sum_i b_i log(prior scale_i) over the top-level latents, plus the base
  double c = m->const_base;
  for (const auto& ts : m->top_scale) c -= (double)b_host[ts.first] * ts.second;
  m->logp_const[which] = c;
  return 0;
}

static const LaneOps* select_ops(arp_model* m, int K_req, int C) {
  const auto* fam = family(m);
  if (!fam) { set_error("model family has no kernels"); return nullptr; }
  if (K_req != 0 && K_req != 1 && K_req != 2 && K_req != 4 && K_req != 8 && K_req != 16) {
    set_error("lanes_per_chain must be 0,1,2,4,8 or 16");
    return nullptr;
  }
  // german credit: the 4-lane instantiation runs its likelihood on the matrix cores and beats the
  // wider ones at every chain count (per workgroup 4x the 8-lane and 17x the 16-lane rate)
  if (K_req == 0 && m->model == ARP_MODEL_GERMAN_CREDIT) K_req = 4;
  // radon: a wider split costs more replicated work than a second wave per SIMD returns (bench.py --chains 8192:
  // 1.21e10 leapfrog-steps/s at 8 lanes per chain, 1.07e10 at 16), so one wave per SIMD is enough
  // time_series likewise: 4 lanes per chain (one wave per SIMD at 16 384 chains) beat 8 and 16 in every form wherever
  // they fill the SIMDs once (round 3 sweep, profiles/r03_time_series_sweep.txt)
  const long long fill = (m->model == ARP_MODEL_RADON || m->model == ARP_MODEL_TIME_SERIES) ? 65536 : 131072;
  // German credit at 4 lanes per chain: the likelihood on bf16 matrix cores with three-piece operands where the data
  // allow it (model_german.h), unless the caller asked for the f32 matrix-core form (arp_model_set_option)
  if (m->model == ARP_MODEL_GERMAN_CREDIT && K_req == 4 && m->german.Xb && m->german_math != 1) return &german_bf3_ops();
  const LaneOps* o = pick(*fam, m->n_groups, K_req, C, m->model != ARP_MODEL_GERMAN_CREDIT, fill,
                          m->model == ARP_MODEL_TIME_SERIES ? 2 : 1);
  if (!o) set_error("no kernel instantiation for this (lanes_per_chain, group count): add <Model>Lane<K, ceil(groups/K)> to the model's inst_*.hip");
  return o;
}

int arp_logp_grad(arp_model* m, int which, const float* x, int n_chains, float* logp, float* grad,
                  int lanes_per_chain, void* stream) {
  if (!m || which < 0 || which > 1 || !x || !logp || !grad || n_chains <= 0) { set_error("arp_logp_grad: bad argument"); return 1; }
  const LaneOps* o = select_ops(m, lanes_per_chain, n_chains);
  if (!o) return 1;
  o->logp_grad(family_args(m), m->dev_ab[which], m->dev_ab[which] + m->D, x, n_chains, m->D, logp, grad,
               (hipStream_t)stream);
  ARP_HIP_OK(hipGetLastError());
  return 0;
}

int arp_transform(arp_model* m, int which, int dir, const float* in, int n_chains, float* out, void* stream) {
  if (!m || which < 0 || which > 1 || !in || !out || n_chains <= 0 || (dir != 0 && dir != 1)) { set_error("arp_transform: bad argument"); return 1; }
  const LaneOps* o = select_ops(m, 0, n_chains);
  if (!o) return 1;
  o->transform(family_args(m), m->dev_ab[which], m->dev_ab[which] + m->D, dir, in, n_chains, m->D, out,
               (hipStream_t)stream);
  ARP_HIP_OK(hipGetLastError());
  return 0;
}

// The step-size recurrences compare log alpha itself with log(target) (kernels.h: adapt_update), which equals TFP's
// min(log alpha, 0) > log(target) only for a target below 1; a rate <= -1 would flip or zero the step.
static int check_adapt(const arp_hmc_config* cfg) {
  if (cfg->adapt_kind == ARP_ADAPT_NONE) return 0;
  if (!(cfg->adapt_target > 0.0f && cfg->adapt_target < 1.0f)) { set_error("adapt_target must lie in (0, 1)"); return 1; }
  if (cfg->adapt_kind == ARP_ADAPT_SIMPLE && !(cfg->adapt_rate > 0.0f)) { set_error("adapt_rate must be positive"); return 1; }
  return 0;
}

// Relay segments (kernels.h: relay_begin; host_common.h: relay_plan): a launch's steps cut into segments that are handed
// from workgroup to workgroup inside the launch, so that a CU that is free takes the next (segment, chain block) in line
// instead of idling behind a slower one (profiles/r05_relay_segments.txt).  This gives the launcher what it needs: one
// zeroed flag word per chain block that belongs to THIS launch alone -- a stream-ordered allocation, given back behind the
// launch (relay_release), so launches of one handle that overlap on different streams never share a word -- and `segs` =
// -1 (the launcher decides with its kernel's occupancy), a forced count (ARP_DEBUG=1 ARP_SEGMENTS=n) or 1 (`allowed` false).
static int relay_prepare(arp_model* m, const arp_hmc_config* cfg, int K, bool allowed, hipStream_t stream, HmcParams* P) {
  P->segs = 1; P->seg_len = cfg->n_steps; P->seg_blocks = 0; P->seg_epoch = 0; P->seg_flags = nullptr;
  P->seg_ctrl = nullptr; P->seg_err_host = nullptr; P->seg_timeout = 6000000000ull; P->seg_fault = 0;
  arp::relay_device_cus() = m->cus;
  if (!allowed || cfg->n_steps < 256) return 0;
  int segs = -1, dbg = 0;
  if (debug_int("ARP_SEGMENTS", &dbg) && dbg >= 1 && dbg <= 64) segs = dbg;
  if (segs == 1) return 0;
  const long long blocks = ((long long)cfg->n_chains * K + kBlock - 1) / kBlock;
  // no kernel gets segments below one round of two workgroups per CU (relay_plan): spare those launches the allocation
  if (segs == -1 && blocks < 2LL * m->cus) return 0;
  // one zeroed flag word per chain block, then the launch's ticket counter and its failure word
  void* flags = nullptr;
  const size_t bytes = ((size_t)blocks + 2) * sizeof(unsigned);
  ARP_HIP_OK(hipMallocAsync(&flags, bytes, stream));
  const hipError_t zeroed = hipMemsetAsync(flags, 0, bytes, stream);
  if (zeroed != hipSuccess) {
    (void)hipFreeAsync(flags, stream);
    ARP_HIP_OK(zeroed);
  }
  P->segs = segs; P->seg_blocks = (int)blocks; P->seg_epoch = 0u; P->seg_flags = (unsigned*)flags;
  P->seg_ctrl = (unsigned*)flags + blocks;
  P->seg_err_host = m->relay_err_dev;
  // test hooks (ARP_DEBUG=1 only): a short time-out, and segments that never raise their flag -- the failure path on demand
  if (debug_int("ARP_RELAY_TIMEOUT_MS", &dbg) && dbg > 0) P->seg_timeout = 100000ull * (unsigned long long)dbg;
  if (debug_int("ARP_RELAY_FAULT", &dbg) && dbg == 1) P->seg_fault = 1;
  return 0;
}
// a relay launch of this handle whose hand-over timed out (kernels.h: relay_begin) since the last look: report it once
static int relay_failed(arp_model* m, const char* where) {
  if (!m || !m->relay_err) return 0;
  if (__atomic_load_n(m->relay_err, __ATOMIC_ACQUIRE) == 0u) return 0;
  __atomic_store_n(m->relay_err, 0u, __ATOMIC_RELEASE);
  set_error(std::string(where) + ": a relay hand-over inside an earlier chain launch of this handle timed out (was the device "
            "taken away for a minute?); that launch left its chains partly advanced -- discard them");
  return 1;
}
static int relay_release(const HmcParams& P, hipStream_t stream) {
  if (P.seg_flags) ARP_HIP_OK(hipFreeAsync(P.seg_flags, stream));
  return 0;
}

static int fill_params(arp_model* m, const arp_hmc_config* cfg, const arp_hmc_io* io, bool need_cache, HmcParams* Pp) {
  if (cfg->n_chains <= 0 || cfg->n_leapfrog <= 0 || cfg->n_steps < 0 || cfg->thin <= 0 || cfg->step_base < 0) {
    set_error("n_chains, n_leapfrog, thin must be positive and n_steps, step_base non-negative");
    return 1;
  }
  if (!io->q || !io->adapt || !io->rng || !io->accept_count || !io->eps0 ||
      (need_cache && (!io->grad || !io->logp))) {
    set_error("q, grad, logp, adapt, rng, accept_count and eps0 are required");
    return 1;
  }
  if (cfg->adapt_kind < ARP_ADAPT_NONE || cfg->adapt_kind > ARP_ADAPT_SIMPLE) { set_error("bad adapt_kind"); return 1; }
  if (check_adapt(cfg)) return 1;
  if (m->D > kMaxD) { set_error("state dimension exceeds the chain kernels' limit (256)"); return 1; }
  HmcParams& P = *Pp;
  P.C = cfg->n_chains; P.L = cfg->n_leapfrog; P.n_steps = cfg->n_steps;
  P.step_base = cfg->step_base; P.chain_offset = cfg->chain_offset; P.seed = cfg->seed;
  P.adapt_kind = cfg->adapt_kind; P.n_adapt = cfg->n_adapt;
  P.adapt_target = cfg->adapt_target; P.adapt_rate = cfg->adapt_rate;
  P.adapt_log_target = logf(cfg->adapt_target > 0.f ? cfg->adapt_target : 1e-30f);
  P.adapt_inv_opr = 1.0f / (1.0f + cfg->adapt_rate);
  P.n_burnin = cfg->n_burnin; P.thin = cfg->thin;
  P.n_samples = (io->trace || io->trace_accept || io->stats || io->rec_accept_count) ? cfg->n_samples : 0;
  if (io->stats && cfg->stats_batch < 1) { set_error("stats_batch must be >= 1 when stats is given"); return 1; }
  P.trace_centered = cfg->trace_centered;
  {
    // result r is taken after transition n = 1 + burnin + r*thin (1-based, global);
    // in-launch step s completes transition step_base + s + 1
    long long first_n = 1 + (long long)cfg->n_burnin;
    long long r0 = 0;
    if (cfg->step_base + 1 > first_n) {
      r0 = (cfg->step_base + 1 - first_n + cfg->thin - 1) / cfg->thin;
      first_n += r0 * cfg->thin;
    }
    long long s0 = first_n - cfg->step_base - 1;
    P.rec_step = s0 < cfg->n_steps ? (int)s0 : -1;
    P.rec_row = (int)(r0 < 0x7fffffff ? r0 : 0x7fffffff);
    P.stats_batch = cfg->stats_batch > 0 ? cfg->stats_batch : 1;
    P.stats_bpos = (int)(r0 % P.stats_batch);
  }
  P.D = m->D;
  P.q = io->q; P.grad = io->grad; P.logp = io->logp; P.adapt = io->adapt;
  P.rng = io->rng; P.accept_count = io->accept_count; P.eps0 = io->eps0;
  P.trace = io->trace; P.trace_accept = io->trace_accept; P.stats = io->stats; P.rec_accept = io->rec_accept_count;
  P.trace_chains = (cfg->trace_chains > 0 && cfg->trace_chains < cfg->n_chains) ? cfg->trace_chains : cfg->n_chains;
  P.L1 = 0; P.adapt1 = nullptr; P.accept_count1 = nullptr; P.eps0_1 = nullptr; P.trace_accept1 = nullptr;
  P.rec_accept1 = nullptr;
  P.segs = 1; P.seg_len = cfg->n_steps; P.seg_blocks = 0; P.seg_epoch = 0; P.seg_flags = nullptr;
  P.seg_ctrl = nullptr; P.seg_err_host = nullptr; P.seg_timeout = 6000000000ull; P.seg_fault = 0;
  return 0;
}

int arp_model_check(arp_model* m) {
  if (!m) { set_error("arp_model_check: null argument"); return 1; }
  return relay_failed(m, "arp_model_check");
}

int arp_hmc_run(arp_model* m, int which, const arp_hmc_config* cfg, const arp_hmc_io* io, void* stream) {
  if (!m || !cfg || !io || which < 0 || which > 1) { set_error("arp_hmc_run: null argument"); return 1; }
  if (relay_failed(m, "arp_hmc_run")) return 1;
  HmcParams P;
  if (fill_params(m, cfg, io, true, &P)) return 1;
  if (cfg->n_steps == 0) return 0;
  const LaneOps* o = select_ops(m, cfg->lanes_per_chain, cfg->n_chains);
  if (!o) return 1;
  auto fn = o->hmc;
  if (m->param_kind[which] == kModeCP && o->hmc_cp) fn = o->hmc_cp;
  if (m->param_kind[which] == kModeNCP && o->hmc_ncp) fn = o->hmc_ncp;
  if (m->param_kind[which] == kModeB1 && o->hmc_b1) fn = o->hmc_b1;
  if (m->param_kind[which] == kModeVIP && o->hmc_vip_pk) fn = o->hmc_vip_pk;
  if (relay_prepare(m, cfg, o->K, true, (hipStream_t)stream, &P)) return 1;
  fn(family_args(m), m->dev_ab[which], m->dev_ab[which] + m->D, P, (hipStream_t)stream);
  const hipError_t launched = hipGetLastError();
  if (relay_release(P, (hipStream_t)stream)) return 1;
  ARP_HIP_OK(launched);
  return 0;
}

int arp_interleaved_run(arp_model* m, const arp_hmc_config* cfg, int n_leapfrog_1,
                        const arp_interleaved_io* io, void* stream) {
  if (!m || !cfg || !io) { set_error("arp_interleaved_run: null argument"); return 1; }
  if (n_leapfrog_1 <= 0 || !io->adapt1 || !io->accept_count1 || !io->eps0_1) {
    set_error("arp_interleaved_run: n_leapfrog_1, adapt1, accept_count1 and eps0_1 are required");
    return 1;
  }
  if (io->k0.grad && !io->k0.logp) {
    // kernels that carry the gradient across the change of coordinates keep BOTH between calls
    set_error("arp_interleaved_run: k0.logp is required whenever k0.grad is given (pass both or neither)");
    return 1;
  }
  if (relay_failed(m, "arp_interleaved_run")) return 1;
  HmcParams P;
  if (fill_params(m, cfg, &io->k0, false, &P)) return 1;
  if (io->trace_accept1 && !P.n_samples) P.n_samples = cfg->n_samples;
  P.L1 = n_leapfrog_1; P.adapt1 = io->adapt1; P.accept_count1 = io->accept_count1;
  P.eps0_1 = io->eps0_1; P.trace_accept1 = io->trace_accept1; P.rec_accept1 = io->rec_accept_count1;
  if (cfg->n_steps == 0) return 0;
  const LaneOps* o = select_ops(m, cfg->lanes_per_chain, cfg->n_chains);
  if (!o) return 1;
  auto fn = o->interleaved;
  if (m->param_kind[0] == kModeCP && m->param_kind[1] == kModeNCP && o->interleaved_cp_ncp) fn = o->interleaved_cp_ncp;
  // (kernels that carry the gradient from step to step need it to travel with the state, as it does between launches)
  if (relay_prepare(m, cfg, o->K, io->k0.grad != nullptr, (hipStream_t)stream, &P)) return 1;
  fn(family_args(m), m->dev_ab[0], m->dev_ab[0] + m->D, m->dev_ab[1], m->dev_ab[1] + m->D, P, (hipStream_t)stream);
  const hipError_t launched = hipGetLastError();
  if (relay_release(P, (hipStream_t)stream)) return 1;
  ARP_HIP_OK(launched);
  return 0;
}

int arp_vi_run(arp_model* m, int which, const arp_vi_config* cfg, const arp_vi_io* io, void* stream) {
  if (!m || !cfg || !io || which < 0 || which > 1) { set_error("arp_vi_run: null argument"); return 1; }
  if (cfg->n_lr <= 0 || cfg->n_steps <= 0 || cfg->n_mc <= 0 || cfg->n_mc > 4096) {
    set_error("arp_vi_run: n_lr, n_steps must be positive and 0 < n_mc <= 4096");
    return 1;
  }
  if (!io->lr || !io->loc || !io->rho || !io->elbo || (cfg->learn_a && !io->w)) {
    set_error("arp_vi_run: lr, loc, rho, elbo (and w when learn_a) are required");
    return 1;
  }
  if (m->D > kViDmax) { set_error("arp_vi_run: model dimension exceeds the VI kernel's limit"); return 1; }
  const auto* fam = family(m);
  if (!fam) { set_error("model family has no kernels"); return 1; }
  // the VI kernel wants the smallest per-lane slice: the widest lanes-per-chain instantiation that has one
  // (german credit: its matrix-core instantiation)
  int Kmax = 0;
  for (const auto& o : *fam) if (o.vi) Kmax = std::max(Kmax, o.K);
  if (m->model == ARP_MODEL_GERMAN_CREDIT) Kmax = 4;
  const LaneOps* o = pick(*fam, m->n_groups, Kmax, 1 << 30, m->model != ARP_MODEL_GERMAN_CREDIT, 131072,
                          m->model == ARP_MODEL_TIME_SERIES ? 2 : 1);
  if (!o || !o->vi) { set_error("no VI kernel instantiation covers this group count"); return 1; }
  // German credit: the row-part lane on bf16 matrix cores with three-piece operands where the data allow it (64-observation
  // tiles), else on f32 matrix cores (128-observation tiles) -- the same choice as the chain kernels make (select_ops)
  const bool german_bf3 = m->model == ARP_MODEL_GERMAN_CREDIT && m->german.Xb && m->german_math != 1;
  if (german_bf3) o = &german_bf3_ops();
  if (m->D > o->vi_dmax) { set_error("arp_vi_run: model dimension exceeds this model's VI kernel instantiation"); return 1; }
  ViParams P;
  P.n_steps = cfg->n_steps; P.n_mc = cfg->n_mc; P.learn_a = cfg->learn_a; P.tied_b = cfg->tied_b; P.a_prior = cfg->a_prior; P.D = m->D;
  P.seed = cfg->seed;
  P.const_base = (float)m->const_base;
  P.n_top = (int)m->top_scale.size();
  for (int k = 0; k < 4; ++k) { P.top_idx[k] = 0; P.top_logscale[k] = 0.f; }
  for (int k = 0; k < P.n_top; ++k) { P.top_idx[k] = m->top_scale[k].first; P.top_logscale[k] = (float)m->top_scale[k].second; }
  P.lr = io->lr; P.loc = io->loc; P.rho = io->rho; P.w = io->w; P.wb = cfg->learn_a ? io->wb : nullptr;
  P.elbo = io->elbo;
  P.prior = cfg->a_prior ? io->prior : nullptr;
  P.a_group = (cfg->learn_a && !cfg->tied_b) ? io->a_group : nullptr;
  P.b_group = (cfg->learn_a && io->wb) ? io->b_group : nullptr;
  // The kernel indexes LDS with these (device) arrays and assumes a group's members sit contiguously behind their
  // leader: check that here, once per fit ([D] ints each).
  for (const int* grp : {P.a_group, P.b_group}) {
    if (!grp) continue;
    std::vector<int> g(m->D);
    ARP_HIP_OK(hipMemcpyAsync(g.data(), grp, m->D * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    ARP_HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    for (int d = 0; d < m->D; ++d) {
      const int l = g[d];
      bool ok = l >= 0 && l <= d && g[l] == l;
      for (int e = l; ok && e <= d; ++e) ok = g[e] == l;     // every element between the leader and d belongs to it
      if (!ok) {
        set_error("arp_vi_run: a_group / b_group must map every element to the first element of a contiguous group (0 <= g[d] <= d, g[g[d]] == g[d])");
        return 1;
      }
    }
  }
  // ---- geometry of the launch (kernels.h: vi_kernel): G sample groups x R row parts per learning rate
  const int B = o->vi_block, K = o->K;
  const int CPW = B / K;                                 // draws a workgroup takes per pass
  const int unit = kViBlock / B;                         // G must be a multiple of this when a lane takes several draws
  int G = (cfg->n_mc + CPW - 1) / CPW;                   // one pass
  int R = 1;
  if (o->vi_parts) {
    // German credit: the observations' 128-row tiles are split too, so that a learning rate's group has about 32
    // workgroups (five learning rates: 160 CUs) and a gradient is two tiles of matrix-core work per wave
    const int tile_obs = german_bf3 ? kBf3Rows : kGermanTileRows;
    const int nt = (m->german.N + tile_obs - 1) / tile_obs;
    R = std::min(nt, std::max(1, 32 / G));
  }
  int dbg = 0;
  if (debug_int("ARP_VI_G", &dbg) && dbg > 0 && !o->vi_parts) G = std::min(G, dbg);      // experiments (ARP_DEBUG=1 only)
  if (debug_int("ARP_VI_R", &dbg) && dbg > 0 && o->vi_parts) R = dbg;
  int occ = o->vi_occ ? o->vi_occ() : 0;
  if (occ <= 0) { set_error("arp_vi_run: the VI kernel does not fit on this device (occupancy query)"); return 1; }
  // every workgroup of a group has to be resident together (they wait for each other twice per step).  One or two per
  // CU is an LDS or register-file limit, which the query gets right; at more than that it can be one workgroup per CU
  // high (MI355X_MICROARCH.md, residency: the scalar-register edge), so one is given away
  const long long capacity = (long long)(occ > 2 ? occ - 1 : occ) * m->cus;
  while ((long long)G * R > capacity && R > 1) R = (R + 1) / 2;
  if ((long long)G * R > capacity && o->vi_parts) {
    set_error("arp_vi_run: n_mc draws of this model do not fit on the device in one pass");   // 4 096 draws: 128 workgroups
    return 1;
  }
  if ((long long)G * R > capacity) G = (int)std::max<long long>(unit, capacity / R / unit * unit);
  if ((long long)G * CPW < cfg->n_mc && G % unit != 0) G = (G + unit - 1) / unit * unit;   // several draws per lane: whole turns
  if ((long long)G * R > capacity) { set_error("arp_vi_run: one learning rate's workgroups do not fit on this device"); return 1; }
  const int GR = G * R;
  const int groups_per_launch = (int)std::max<long long>(1, std::min<long long>(cfg->n_lr, capacity / GR));
  const int nq = cfg->learn_a ? 4 : 2;
  const int Tp = (nq * m->D + 1 + 15) & ~15;
  const size_t need = GR > 1 ? 256 + groups_per_launch * vi_xch_group_granules(GR, Tp) * 8 : 256;
  if (m->vi_ws_bytes < need) {
    if (m->vi_ws) { ARP_HIP_OK(hipStreamSynchronize((hipStream_t)stream)); (void)hipFree(m->vi_ws); m->vi_ws = nullptr; m->vi_ws_bytes = 0; }
    ARP_HIP_OK(hipMalloc(&m->vi_ws, need));
    m->vi_ws_bytes = need;
  }
  P.G = G; P.R = R; P.xch_tp = Tp;
  P.err = (int*)m->vi_ws;
  P.xch = GR > 1 ? (unsigned long long*)((char*)m->vi_ws + 256) : nullptr;
  g_vi_geometry = {B, G, R, groups_per_launch, (int)std::min<long long>((long long)cfg->n_lr * GR, capacity), occ};
  // A group's workgroups wait for each other, so all of them must be resident.  Cooperative launch (the default where the
  // device has it): the runtime checks the grid against the device's capacity and serialises such launches of the process.
  // Plain launch: the occupancy arithmetic above plus one such launch at a time in this process (the mutex).  Neither holds
  // against kernels of other queues or processes: see the retry below.
  const bool coop = GR > 1 && (m->vi_launch == 2 || (m->vi_launch == 0 && m->coop_ok));
  if (m->vi_launch == 2 && !m->coop_ok) { set_error("arp_vi_run: vi_launch=cooperative but the device does not support cooperative launches"); return 1; }
  std::unique_lock<std::mutex> one_at_a_time(g_vi_launch_mutex, std::defer_lock);
  if (GR > 1 && !coop) one_at_a_time.lock();
  // A launch whose hand-offs ran into their bound (the device was shared with kernels of other queues or processes for
  // that long: no launch mode guarantees residency against THOSE) is not an error yet: the parameters it started from are
  // kept, and it is taken again with four times the bound -- 2 s, 8 s, 32 s -- before the call gives up.  A fit under
  // contention is slower, not failed; an undisturbed one never takes the second launch.
  const size_t rowf = (size_t)m->D;
  const int n_keep = 2 + (io->w ? 1 : 0) + (io->wb && cfg->learn_a ? 1 : 0);
  if (GR > 1 && m->vi_snap_floats < (size_t)n_keep * groups_per_launch * rowf) {
    if (m->vi_snap) { ARP_HIP_OK(hipStreamSynchronize((hipStream_t)stream)); (void)hipFree(m->vi_snap); m->vi_snap = nullptr; m->vi_snap_floats = 0; }
    ARP_HIP_OK(hipMalloc((void**)&m->vi_snap, (size_t)n_keep * groups_per_launch * rowf * sizeof(float)));
    m->vi_snap_floats = (size_t)n_keep * groups_per_launch * rowf;
  }
  int fault_attempts = 0;               // test hook (ARP_DEBUG=1): treat the first n launches of every chunk as timed out
  if (debug_int("ARP_VI_FAULT_ATTEMPTS", &dbg) && dbg > 0) fault_attempts = dbg;
  g_vi_attempts = 0;
  for (int lr0 = 0; lr0 < cfg->n_lr; lr0 += groups_per_launch) {
    const int ng = std::min(groups_per_launch, cfg->n_lr - lr0);
    float* rows[4] = {io->loc, io->rho, io->w, (io->wb && cfg->learn_a) ? io->wb : nullptr};
    const size_t chunk = (size_t)ng * rowf;
    if (GR > 1) {
      size_t k = 0;
      for (float* r : rows)
        if (r) ARP_HIP_OK(hipMemcpyAsync(m->vi_snap + (k++) * chunk, r + (size_t)lr0 * rowf, chunk * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    P.lr0 = lr0;
    P.spin_ticks = kViSpinTicks;
    for (int attempt = 0;; ++attempt) {
      if (attempt > 0) {
        size_t k = 0;
        for (float* r : rows)
          if (r) ARP_HIP_OK(hipMemcpyAsync(r + (size_t)lr0 * rowf, m->vi_snap + (k++) * chunk, chunk * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        P.spin_ticks *= 4;
      }
      // every polled word starts at zero (epochs start at 1): the flag and this launch's granules
      ARP_HIP_OK(hipMemsetAsync(m->vi_ws, 0, need, (hipStream_t)stream));
      hipError_t launched = o->vi(family_args(m), m->dev_ab[which], m->dev_ab[which] + m->D, P, ng, coop, (hipStream_t)stream);
      if (coop && m->vi_launch == 0 && (launched == hipErrorCooperativeLaunchTooLarge || launched == hipErrorNotSupported)) {
        // "auto" only: the runtime counts co-residency more strictly than the occupancy query above (or lacks the feature
        // after all) -- take the plain launch, one at a time per process, as round 5 did
        (void)hipGetLastError();
        if (!one_at_a_time.owns_lock()) one_at_a_time.lock();
        launched = o->vi(family_args(m), m->dev_ab[which], m->dev_ab[which] + m->D, P, ng, false, (hipStream_t)stream);
      }
      ARP_HIP_OK(launched);
      g_vi_attempts = std::max(g_vi_attempts, attempt + 1);
      if (GR <= 1) break;
      // the hand-offs' waits are bounded: a group that was not resident together reports it here instead of hanging
      int err = 0;
      ARP_HIP_OK(hipMemcpyAsync(&err, m->vi_ws, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
      ARP_HIP_OK(hipStreamSynchronize((hipStream_t)stream));
      if (attempt < fault_attempts) err = 1;
      if (!err) break;
      if (attempt == 2) {
        set_error("arp_vi_run: a hand-off between the workgroups of a learning rate timed out three times (2 s, 8 s, 32 s: the "
                  "group was never resident together -- is another process holding the device?)");
        return 1;
      }
    }
  }
  return 0;
}

int arp_vi_attempts(int32_t* out1) {
  if (!out1) { set_error("arp_vi_attempts: null argument"); return 1; }
  *out1 = g_vi_attempts;
  return 0;
}

int arp_relay_geometry(int32_t* out3) {
  if (!out3) { set_error("arp_relay_geometry: null argument"); return 1; }
  for (int i = 0; i < 3; ++i) out3[i] = relay_last().v[i];
  return 0;
}

int arp_vi_geometry(int32_t* out6) {
  if (!out6) { set_error("arp_vi_geometry: null argument"); return 1; }
  for (int i = 0; i < 6; ++i) out6[i] = g_vi_geometry.v[i];
  return 0;
}

int arp_adapt_probe(const arp_hmc_config* cfg, const float* log_accept, int n, float* adapt, float* kappa_out,
                    void* stream) {
  if (!cfg || !log_accept || !adapt || n <= 0 || cfg->n_steps < 0 || cfg->step_base < 0) {
    set_error("arp_adapt_probe: bad argument");
    return 1;
  }
  if (cfg->adapt_kind < ARP_ADAPT_NONE || cfg->adapt_kind > ARP_ADAPT_SIMPLE) { set_error("bad adapt_kind"); return 1; }
  if (check_adapt(cfg)) return 1;
  HmcParams P{};
  P.n_steps = cfg->n_steps; P.step_base = cfg->step_base;
  P.adapt_kind = cfg->adapt_kind; P.n_adapt = cfg->n_adapt;
  P.adapt_target = cfg->adapt_target; P.adapt_rate = cfg->adapt_rate;
  P.adapt_log_target = logf(cfg->adapt_target > 0.f ? cfg->adapt_target : 1e-30f);
  P.adapt_inv_opr = 1.0f / (1.0f + cfg->adapt_rate);
  hipLaunchKernelGGL(adapt_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, log_accept, n, adapt,
                     kappa_out);
  ARP_HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
