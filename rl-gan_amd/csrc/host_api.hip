// Host-only entry points: geometry of the device images, canonical<->fragment index maps,
// error strings.  No GPU work.
#include "rlgan_device.h"
#include "rlgan_host.h"

using namespace rlgan;

namespace {

struct MapOut { int32_t* f; int32_t* b; int32_t* g; int p; const Geo* geo; };

// one weight block of layer l: canonical rows [0,nrows) sit at operand rows k0.. of the layer
void emit_weight(MapOut& m, int l, int k0, int nrows) {
  const Geo& g = *m.geo;
  for (int k = 0; k < nrows; ++k)
    for (int n = 0; n < g.N[l]; ++n, ++m.p) {
      m.f[m.p] = wf_offset(g, l, k0 + k, n);
      if (m.b) m.b[m.p] = wb_offset(g, l, k0 + k, n);
      m.g[m.p] = g.goff[l] + (k0 + k) * g.np[l] + n;
    }
}
void emit_bias(MapOut& m, int l) {
  const Geo& g = *m.geo;
  for (int n = 0; n < g.N[l]; ++n, ++m.p) {
    m.f[m.p] = wf_offset(g, l, g.K[l], n);
    if (m.b) m.b[m.p] = -1;
    m.g[m.p] = g.goff[l] + g.K[l] * g.np[l] + n;
  }
}

}  // namespace

extern "C" {

int rlgan_abi_version(void) { return RLGAN_ABI_VERSION; }

const char* rlgan_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case RLGAN_E_ARG: return "null pointer or negative size";
    case RLGAN_E_DIMS: return "state/action dimensions outside what the kernels are built for (2*s_dim+a_dim <= 16)";
    case RLGAN_E_KIND: return "unknown network kind";
    case RLGAN_E_NOTFULL: return "Memory has not been fulfilled";
    case RLGAN_E_ALIGN: return "row count or leading dimension not as documented";
    case RLGAN_E_SCRATCH: return "size beyond the fixed device scratch of the allocation-free entry point (rlgan_column_sums_noalloc: n_cols <= 256)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

int rlgan_geometry(int kind, int s_dim, int a_dim, rlgan_geom_t* out) {
  if (!out) return RLGAN_E_ARG;
  if (kind < 0 || kind > NET_PCRITIC) return RLGAN_E_KIND;
  Geo g;
  if (!make_geo(kind, s_dim, a_dim, g)) return RLGAN_E_DIMS;
  out->n_params = g.n_params; out->wf_floats = g.wf_floats; out->wb_floats = g.wb_floats;
  out->gpad_floats = g.gpad_floats; out->k0 = g.K0; out->n3 = g.N3; out->n_out = g.NOUT; out->ld3 = g.ld3;
  return 0;
}

// Tensor order = TensorFlow variable creation order (DDPG.py:80-93,155-171; Curious_GAN.py:52-81,
// 114-134).  Input-tile column order: actor [s], critic/generator [s, a], discriminator [s, x, a].
int rlgan_build_maps(int kind, int s_dim, int a_dim, int32_t* map_f, int32_t* map_b, int32_t* map_g) {
  if (!map_f || !map_g) return RLGAN_E_ARG;
  if (kind < 0 || kind > NET_PCRITIC) return RLGAN_E_KIND;
  Geo g;
  if (!make_geo(kind, s_dim, a_dim, g)) return RLGAN_E_DIMS;
  MapOut m{map_f, map_b, map_g, 0, &g};
  const int S = s_dim, A = a_dim;
  switch (kind) {
    case NET_PACTOR:
    case NET_PCRITIC: emit_weight(m, 0, 0, g.K0); break;                               // one [K0, 200] block: features, then action columns
    case NET_ACTOR:  emit_weight(m, 0, 0, S); break;                                   // l1/kernel
    case NET_CRITIC:
    case NET_GEN:    emit_weight(m, 0, 0, S); emit_weight(m, 0, S, A); break;           // w1_s, w1_a
    default:         emit_weight(m, 0, S, S); emit_weight(m, 0, 0, S); emit_weight(m, 0, 2 * S, A); break;  // w1_G, w1_s, w1_a
  }
  emit_bias(m, 0);
  for (int l = 1; l < 4; ++l) { emit_weight(m, l, 0, g.K[l]); emit_bias(m, l); }
  return m.p == g.n_params ? 0 : RLGAN_E_DIMS;
}

}  // extern "C"
