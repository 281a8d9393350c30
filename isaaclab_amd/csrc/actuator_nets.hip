// ActuatorNetLSTM / ActuatorNetMLP on the matrix cores (v_mfma_f32_16x16x4_f32: exact-f32 MFMA).
// Reference: LAB/actuators/actuator_net.py:30-188.
//
// Mapping (one wave = 16 actuator rows at a time, rows = N*J):
//   D tile (16x16) = W (16 outputs x 4 k) * X (4 k x 16 rows); lane l -> column (row of the batch)
//   l & 15, quad gq = l >> 4.  The weight rows are permuted on the host (mdp_net_pack) so that
//     * LSTM tile t, D row gq*4 + r  = gate r (i,f,g,o) of hidden unit u = gq*(H/4) + t
//       -> all four gates of a unit land in ONE lane: the cell update needs no cross-lane traffic;
//     * the K order of the next product follows the D layout of the previous one
//       (k-slot gq of k-step (t, r) = value held in register r of tile t by quad gq)
//       -> layer outputs feed the next MFMA as the B operand straight from registers.
//   Weights sit in LDS as ready-made A fragments (64 floats, one per lane, conflict-free ds_read_b32).
// HBM traffic per row and sub-step: inputs 12 B, (h, c) 2*2*L*H*4 B, outputs 8 B -> HBM-bound
// (SURVEY.md section 8d); the MFMA pipe is there to keep the VALU free for the gate transcendentals.
#include "mdp_device.hpp"
#include "mdp_host.hpp"
#include "stamps_device.hpp"
#include "joint_action_device.hpp"
#include "scene_update_device.hpp"
#include "raycaster_update_device.hpp"
#include "raycaster_update_host.hpp"

#include <cstdlib>
#include <vector>

using namespace mdp;

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAX_MLP_LAYERS = 4;
constexpr int MLP_NT = 2;  // tiles per MLP layer -> widths <= 32

struct NetLayout {
    // offsets in floats into the packed blob
    int lstm_w[4];     // per layer: fragments [(kstep * TPL + t) * 64]
    int lstm_b[4];     // per layer: [(t * 4 + r) * 64]
    int lstm_ksteps_in[4];
    int mlp_w[MAX_MLP_LAYERS];  // [((tin * 4 + r) * MLP_NT + tout) * 64] (first layer after an LSTM: [(s * MLP_NT + tout) * 64])
    int mlp_b[MAX_MLP_LAYERS];  // [(tout * 4 + r) * 64]
    int mlp_ksteps[MAX_MLP_LAYERS];
    int total;         // floats of the fragment tables above (what the LDS kernels stage)
    int reg8;          // k_lstm8_reg's copy of its fragments, four per lane and load (reg8_sources), or -1
    int total_all;     // size of the blob: total + the reg8 section
};

// Networks k_lstm8_reg takes (the ANYdrive shape): hidden 8, <= 2 layers, Linear(8 -> 1) decoder.
bool reg8_shape(const mdp_net_desc_t& d) { return d.lstm_layers >= 1 && d.lstm_layers <= 2 && d.lstm_hidden == 8 && d.mlp_layers == 1; }
// The fragments a lane of k_lstm8_reg holds, in the order it unpacks them: per layer the weight fragments (k-step major)
// and the eight bias fragments, then the decoder's two weight and four bias fragments.  src[i] = offset of fragment i in
// the tables above; returns their number (a multiple of 4: 20 for one layer, 36 for two).
int reg8_sources(const mdp_net_desc_t& d, const NetLayout& L, int* src) {
    constexpr int TPL = 2;
    int n = 0;
    for (int l = 0; l < d.lstm_layers; ++l) {
        const int nk = (l == 0 ? 1 : TPL) + TPL;
        for (int k = 0; k < nk; ++k)
            for (int t = 0; t < TPL; ++t) src[n++] = L.lstm_w[l] + (k * TPL + t) * 64;
        for (int i = 0; i < 4 * TPL; ++i) src[n++] = L.lstm_b[l] + i * 64;
    }
    for (int s2 = 0; s2 < TPL; ++s2) src[n++] = L.mlp_w[0] + (s2 * 2) * 64;   // (s * MLP_NT) * 64, tile 0
    for (int r = 0; r < 4; ++r) src[n++] = L.mlp_b[0] + r * 64;
    return n;
}

int tiles_for(int width) { return (width + 15) / 16; }

int make_layout(const mdp_net_desc_t& d, NetLayout& L) {
    int off = 0;
    int H = d.lstm_hidden, TPL = H / 4;
    for (int l = 0; l < d.lstm_layers; ++l) {
        int ks_in = (l == 0) ? (d.lstm_input + 3) / 4 : TPL;
        L.lstm_ksteps_in[l] = ks_in;
        L.lstm_w[l] = off;
        off += (ks_in + TPL) * TPL * 64;
        L.lstm_b[l] = off;
        off += TPL * 4 * 64;
    }
    for (int l = 0; l < d.mlp_layers; ++l) {
        int ks;
        if (l == 0) ks = d.lstm_layers > 0 ? TPL : (d.mlp_dims[0] + 3) / 4;
        else ks = tiles_for(d.mlp_dims[l]) * 4;
        L.mlp_ksteps[l] = ks;
        L.mlp_w[l] = off;
        off += ks * MLP_NT * 64;
        L.mlp_b[l] = off;
        off += MLP_NT * 4 * 64;
    }
    L.total = off;
    L.reg8 = -1;
    if (reg8_shape(d) && d.lstm_input <= 4) {
        int src[40];
        L.reg8 = off;
        off += reg8_sources(d, L, src) * 64;
    }
    L.total_all = off;
    return off;
}

int validate(const mdp_net_desc_t* d) {
    MDP_REQUIRE(d, "null net desc");
    MDP_REQUIRE(d->lstm_layers >= 0 && d->lstm_layers <= 4, "lstm_layers must be in [0,4]");
    if (d->lstm_layers > 0) {
        MDP_REQUIRE(d->lstm_hidden == 8 || d->lstm_hidden == 16 || d->lstm_hidden == 32,
                    "lstm hidden size %d unsupported (8, 16, 32)", d->lstm_hidden);
        MDP_REQUIRE(d->lstm_input >= 1 && d->lstm_input <= 4, "lstm input size %d unsupported (<= 4)", d->lstm_input);
    }
    MDP_REQUIRE(d->mlp_layers >= 1 && d->mlp_layers <= MAX_MLP_LAYERS, "mlp_layers must be in [1,%d]", MAX_MLP_LAYERS);
    for (int l = 0; l <= d->mlp_layers; ++l)
        MDP_REQUIRE(d->mlp_dims[l] >= 1 && d->mlp_dims[l] <= 16 * MLP_NT, "mlp width %d unsupported (<= %d)",
                    d->mlp_dims[l], 16 * MLP_NT);
    MDP_REQUIRE(d->mlp_dims[d->mlp_layers] == 1, "network output width must be 1");
    if (d->lstm_layers > 0) MDP_REQUIRE(d->mlp_dims[0] == d->lstm_hidden, "decoder input must equal lstm hidden");
    for (int l = 0; l < d->mlp_layers; ++l) MDP_REQUIRE(d->mlp_act[l] >= 0 && d->mlp_act[l] <= 5, "bad activation id");
    return MDP_OK;
}

MDP_DEV float act_fn(float x, int a) {
    switch (a) {
        case 1: return x > 0.0f ? x : (expf(x) - 1.0f);          // elu (alpha 1)
        case 2: return x > 0.0f ? x : 0.0f;                      // relu
        case 3: return tanhf(x);
        case 4: return x / (1.0f + fabsf(x));                    // softsign
        case 5: return 1.0f / (1.0f + expf(-x));                 // sigmoid
        default: return x;
    }
}
// LSTM gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, 1 ulp each).  The gate math is
// what bounds the LSTM kernels (20 sigmoid/tanh per lane and row group; libm expf / tanhf cost 15-40 VALU
// instructions each, these 5-12), and the error stays two orders below the 1e-5 parity tolerance:
//   sigmoid: relative error <= ~(|x| log2e + 2) * 2^-24  (1e-6 at |x| = 10);
//   tanh, |x| >= 0.125: (1 - t) / (1 + t) with t = exp(-2|x|), cancellation bounded by 1 / (1 - e^-0.25) = 4.5;
//   tanh, |x| <  0.125: odd Taylor polynomial to x^9 (truncation < 2e-11 relative).
MDP_DEV float sigmoidf_(float x) {
    float t = __builtin_amdgcn_exp2f(x * -1.44269504088896340736f);
    return __builtin_amdgcn_rcpf(1.0f + t);
}
MDP_DEV float tanhf_(float x) {
    float ax = fabsf(x);
    float t = __builtin_amdgcn_exp2f(ax * -2.88539008177792681472f);
    float big = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
    float x2 = x * x;
    float small = ax * (1.0f + x2 * (-0.33333333333f + x2 * (0.13333333333f + x2 * (-0.05396825397f + x2 * 0.02186948854f))));
    float r = ax < 0.125f ? small : big;
    return copysignf(r, x);
}

struct NetDev {
    mdp_net_desc_t d;
    NetLayout L;
};

MDP_DEV f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

MDP_DEV f32x4 load_bias(const float* lds, int off, int t, int lane) {
    f32x4 v;
    v[0] = lds[off + (t * 4 + 0) * 64 + lane];
    v[1] = lds[off + (t * 4 + 1) * 64 + lane];
    v[2] = lds[off + (t * 4 + 2) * 64 + lane];
    v[3] = lds[off + (t * 4 + 3) * 64 + lane];
    return v;
}

// Runs the MLP part.  in_first[]: B operands of the first layer's k-steps (count = mlp_ksteps[0] <= 8).
// Returns the scalar network output in lanes with gq == 0 (D tile 0, row 0).
MDP_DEV float run_mlp(const NetDev& net, const float* lds, int lane, const float (&in_first)[8]) {
    float cur[MLP_NT * 4];  // layer outputs: register r of tile t -> cur[t*4 + r]
#pragma unroll
    for (int i = 0; i < MLP_NT * 4; ++i) cur[i] = 0.0f;
#pragma unroll
    for (int l = 0; l < MAX_MLP_LAYERS; ++l) {
        if (l < net.d.mlp_layers) {
            int nt_out = (net.d.mlp_dims[l + 1] + 15) / 16;
            f32x4 acc[MLP_NT];
#pragma unroll
            for (int t = 0; t < MLP_NT; ++t) acc[t] = load_bias(lds, net.L.mlp_b[l], t, lane);
            int ks = net.L.mlp_ksteps[l];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s < ks) {
                    float b = (l == 0) ? in_first[s] : cur[s];
#pragma unroll
                    for (int t = 0; t < MLP_NT; ++t)
                        if (t < nt_out) acc[t] = mfma4(lds[net.L.mlp_w[l] + (s * MLP_NT + t) * 64 + lane], b, acc[t]);
                }
            }
            int a = net.d.mlp_act[l];
#pragma unroll
            for (int t = 0; t < MLP_NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) cur[t * 4 + r] = act_fn(acc[t][r], a);
        }
    }
    return cur[0];
}

MDP_DEV void dc_clip_store(float tau, float qd, int64_t row, const float* __restrict__ eff_lim,
                           const float* __restrict__ vel_lim, float sat, float* __restrict__ computed,
                           float* __restrict__ applied) {
    float lim = eff_lim[row], vl = vel_lim[row];
    float mx = tclamp(sat * (1.0f - qd / vl), 0.0f, lim);
    float mn = tclamp(sat * (-1.0f - qd / vl), -lim, 0.0f);
    computed[row] = tau;
    applied[row] = tclamp(tau, mn, mx);
}

// Blocks >= nb_act carry the scene update of the frame this launch reads (see k_actuator_pd).
template <int H>
__global__ void __launch_bounds__(256)
k_lstm(mdp_scene_update_t upd, float dt, unsigned nb_act, unsigned nb_root, unsigned nb_joint, unsigned nb_contact, int upd_envs,
       mdp_joint_action_t jact, NetDev net, const float* __restrict__ packed, int64_t rows, int J,
       const float* __restrict__ q_des, int ld_q_des, const float* __restrict__ q, int ld_q,
       const float* __restrict__ qd, int ld_qd, float* __restrict__ hs, float* __restrict__ cs,
       const float* __restrict__ eff_lim, const float* __restrict__ vel_lim, float sat,
       float* __restrict__ computed, float* __restrict__ applied, const float* __restrict__ gate) {
    constexpr int TPL = H / 4;
    extern __shared__ __align__(16) float lds[];
    if (gate != nullptr && *gate <= 0.0f) return;      // mdp_actuator_lstm_compute_gated: nothing happens this launch
    // the carried scene update takes the FIRST blocks of the grid: its chain of dependent loads starts with the launch
    // instead of behind the actuator blocks
    const unsigned nb_upd = gridDim.x - nb_act;
    if (blockIdx.x < nb_upd) {
        scene_update_carried(upd, dt, blockIdx.x, nb_root, nb_joint, nb_contact, upd_envs);
        return;
    }
    const unsigned bid = blockIdx.x - nb_upd;
    int lane = threadIdx.x & 63;
    int gq = lane >> 4, col = lane & 15;
    int64_t wave = (int64_t)bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int64_t nwaves = (int64_t)nb_act * (blockDim.x >> 6);
    int64_t ngroups = (rows + 15) / 16;
    // Software pipeline: the global loads of a group (inputs and the (h, c) slices of every layer) are issued before
    // the weights are staged / before the previous group's gates are evaluated, so the one dependent HBM round trip
    // of the row overlaps the LDS staging instead of following it.  (The kernel is latency-bound: one group per wave
    // at 4096 envs.)
    constexpr int LMAX = 4;
    float x0 = 0.0f, x1 = 0.0f;
    float hreg[LMAX][TPL], creg[LMAX][TPL];
    int64_t rr = 0;
    bool valid = false;
    auto load_group = [&](int64_t g) {
        int64_t row = g * 16 + col;
        valid = row < rows;
        rr = valid ? row : rows - 1;
        int64_t e = rr / J;
        int j = (int)(rr - e * J);
        // actuator_net.py:80-81 (jact.actions_in: this launch also processes the action of the step)
        float des = jact.actions_in ? joint_action_element(jact, e, j, J, valid && gq == 0) : q_des[e * ld_q_des + j];
        x0 = des - q[e * ld_q + j];
        x1 = qd[e * ld_qd + j];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < net.d.lstm_layers) {
                int64_t sbase = ((int64_t)l * rows + rr) * H + gq * TPL;
#pragma unroll
                for (int s = 0; s < TPL; ++s) { hreg[l][s] = hs[sbase + s]; creg[l][s] = cs[sbase + s]; }
            }
        }
    };
    int64_t g = wave;
    if (g < ngroups) load_group(g);
    {
        // weights -> LDS, 128-bit copies (the blob is 256-byte aligned: every offset is a multiple of 64 floats)
        const float4* src = reinterpret_cast<const float4*>(packed);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = threadIdx.x; i < net.L.total / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    while (g < ngroups) {
        const float cx0 = x0, cx1 = x1;
        const bool cvalid = valid;
        const int64_t crr = rr;
        float hl_all[LMAX][TPL], cl_all[LMAX][TPL];
#pragma unroll
        for (int l = 0; l < LMAX; ++l)
#pragma unroll
            for (int s = 0; s < TPL; ++s) { hl_all[l][s] = hreg[l][s]; cl_all[l][s] = creg[l][s]; }
        int64_t gnext = g + nwaves;
        if (gnext < ngroups) load_group(gnext);
        float xin = gq == 0 ? cx0 : (gq == 1 ? cx1 : 0.0f);
        float prev[TPL];
#pragma unroll
        for (int s = 0; s < TPL; ++s) prev[s] = 0.0f;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < net.d.lstm_layers) {
                f32x4 acc[TPL];
#pragma unroll
                for (int t = 0; t < TPL; ++t) acc[t] = load_bias(lds, net.L.lstm_b[l], t, lane);
                const float* w = lds + net.L.lstm_w[l];
                int ks_in = net.L.lstm_ksteps_in[l];
                if (l == 0) {
#pragma unroll
                    for (int t = 0; t < TPL; ++t) acc[t] = mfma4(w[t * 64 + lane], xin, acc[t]);
                } else {
#pragma unroll
                    for (int s = 0; s < TPL; ++s)
#pragma unroll
                        for (int t = 0; t < TPL; ++t) acc[t] = mfma4(w[(s * TPL + t) * 64 + lane], prev[s], acc[t]);
                }
                // recurrent part: this lane holds units gq*TPL .. gq*TPL+TPL-1 of row `col`
                int64_t sbase = ((int64_t)l * rows + crr) * H + gq * TPL;
#pragma unroll
                for (int s = 0; s < TPL; ++s)
#pragma unroll
                    for (int t = 0; t < TPL; ++t)
                        acc[t] = mfma4(w[((ks_in + s) * TPL + t) * 64 + lane], hl_all[l][s], acc[t]);
#pragma unroll
                for (int t = 0; t < TPL; ++t) {
                    float ig = sigmoidf_(acc[t][0]);
                    float fg = sigmoidf_(acc[t][1]);
                    float gg = tanhf_(acc[t][2]);
                    float og = sigmoidf_(acc[t][3]);
                    float cn = fg * cl_all[l][t] + ig * gg;
                    float hn = og * tanhf_(cn);
                    prev[t] = hn;
                    if (cvalid) { hs[sbase + t] = hn; cs[sbase + t] = cn; }
                }
            }
        }
        float in_first[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) in_first[s] = s < TPL ? prev[s < TPL ? s : 0] : 0.0f;
        float tau = run_mlp(net, lds, lane, in_first);
        if (cvalid && gq == 0) dc_clip_store(tau, cx1, crr, eff_lim, vel_lim, sat, computed, applied);
        g = gnext;
    }
}

// The ANYdrive shape (hidden 8, <= 2 layers, Linear(8 -> 1) decoder): every weight fragment a lane needs fits in ~40
// registers, so they are loaded straight from global memory (L2-resident, coalesced 256-B rows) together with the
// row's inputs -- ONE round trip, no LDS staging, no block barrier, no LDS read in front of each MFMA.  The generic
// kernel above spends 63 % of its wave cycles parked on s_waitcnt (SQ_WAIT_ANY) for exactly those.  Same MFMA / gate
// order as k_lstm: bit-identical outputs.
// RECAST (mdp_actuator_lstm_compute_gated_with_recast): the two launches that follow a post-physics launch with masked
// resets -- RayCaster.reset + recast of the reset envs (mdp_raycaster_update_flagged) and the reference's conditional
// write_data_to_sim (mdp_actuator_lstm_compute_gated) -- as ONE: the first nb_cast blocks are recast blocks (theirs is the
// longer chain), the actuator blocks follow.  Neither role reads what the other writes, both do nothing when no env
// was reset (*gate = number of reset envs), and on two streams their hand-over costs what the overlap gains
// (masked-reset step: 106.2 us on two streams, 107.1 one after the other, 103.1 as one launch).  The instantiations
// without RECAST contain none of it (cast / nb_cast are unused arguments there).
template <int NL, bool SLICE, bool RECAST>
__global__ void __launch_bounds__(256)
k_lstm8_reg(RaycasterUpdateArgs cast, unsigned nb_cast, mdp_scene_update_t upd, float dt, unsigned nb_act, unsigned nb_root, unsigned nb_joint, unsigned nb_contact, int upd_envs,
            mdp_joint_action_t jact, NetDev net, const float* __restrict__ packed, int64_t rows, int J,
            const float* __restrict__ q_des, int ld_q_des, const float* __restrict__ q, int ld_q,
            const float* __restrict__ qd, int ld_qd, float* __restrict__ hs, float* __restrict__ cs,
            const float* __restrict__ eff_lim, const float* __restrict__ vel_lim, float sat,
            float* __restrict__ computed, float* __restrict__ applied, const float* __restrict__ gate) {
    constexpr int H = 8, TPL = 2;
    // (recast blocks do not wait for the gate: their own flags decide, as in mdp_raycaster_update_flagged -- with no env
    // reset there is none to cast -- and the gate word would be one more round trip at the head of the launch's longest
    // chain)
    if (RECAST) {
        if (blockIdx.x < nb_cast) {
            raycaster_update_flagged_blocks<2>(cast, blockIdx.x);
            return;
        }
    }
    if (gate != nullptr && *gate <= 0.0f) return;      // mdp_actuator_lstm_compute_gated: nothing happens this launch
    // the carried scene update takes the FIRST blocks of the grid: its chain of dependent loads starts with the launch
    // instead of behind the actuator blocks
    const unsigned nb_upd = RECAST ? nb_cast : gridDim.x - nb_act;
    if (!RECAST && blockIdx.x < nb_upd) {
        MDP_STAMP(0, 1);
        scene_update_carried(upd, dt, blockIdx.x, nb_root, nb_joint, nb_contact, upd_envs);
        MDP_STAMP(1);
        return;
    }
    MDP_STAMP(0, 2);
    const unsigned bid = blockIdx.x - nb_upd;
    // upd_envs < 0: no update blocks -- every actuator block carries the update of a slice of -upd_envs envs itself: its
    // requests go out with the rows' own, its results are stored while the gates are evaluated, and the per-env words
    // are committed behind the block's only barrier at the very end (scene_update_device.hpp: SceneUpdateSlice)
    SceneUpdateSlice us;
    const int us_e0 = SLICE && upd_envs < 0 ? (int)bid * -upd_envs : 0;
    const int us_e1 = SLICE && upd_envs < 0 ? (us_e0 - upd_envs < upd.num_envs ? us_e0 - upd_envs : upd.num_envs) : 0;
    const bool has_slice = SLICE && us_e0 < us_e1;       // block-uniform
    if (has_slice) scene_update_slice_request(upd, us_e0, us_e1, us);
    int lane = threadIdx.x & 63;
    int gq = lane >> 4, col = lane & 15;
    int64_t wave = (int64_t)bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int64_t nwaves = (int64_t)nb_act * (blockDim.x >> 6);
    int64_t ngroups = (rows + 15) / 16;
    float x0 = 0.0f, x1 = 0.0f, lim_r = 0.0f, vl_r = 1.0f;
    float hreg[NL][TPL], creg[NL][TPL];
    int64_t rr = 0;
    bool valid = false;
    // every global input of a row, the limits of the final clip included, is requested in one go (rows < 2^31 here:
    // 32-bit index arithmetic -- a 64-bit division costs more than the whole gate evaluation)
    auto load_group = [&](int64_t g) {
        int64_t row = g * 16 + col;
        valid = row < rows;
        rr = valid ? row : rows - 1;
        const unsigned ur = (unsigned)rr;
        const unsigned ue = ur / (unsigned)J;
        int64_t e = ue;
        int j = (int)(ur - ue * (unsigned)J);
        lim_r = eff_lim[rr];
        vl_r = vel_lim[rr];
        // actuator_net.py:80-81 (jact.actions_in: this launch also processes the action of the step)
        float des = jact.actions_in ? joint_action_element(jact, e, j, J, valid && gq == 0) : q_des[e * ld_q_des + j];
        x0 = des - q[e * ld_q + j];
        x1 = qd[e * ld_qd + j];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            int64_t sbase = ((int64_t)l * rows + rr) * H + gq * TPL;
#pragma unroll
            for (int s = 0; s < TPL; ++s) { hreg[l][s] = hs[sbase + s]; creg[l][s] = cs[sbase + s]; }
        }
    };
    int64_t g = wave;
    if (g < ngroups) load_group(g);
    // weight / bias fragments of this lane (layer 0 has one input k-step, deeper layers TPL): NF numbers, read four at
    // a time from the blob's reg8 section (mdp_net_pack lays them out per lane in exactly this order) -- 9 vector-memory
    // instructions instead of 36: the launch's busiest unit is the texture-address path, ~15 cycles per wave and
    // memory instruction whatever its width (TA_TA_BUSY / TA_TOTAL_WAVEFRONTS), and these loads were 60 % of them
    constexpr int NF = NL == 2 ? 36 : 20;
    float frag[NF];
    {
        const float4* p8 = reinterpret_cast<const float4*>(packed + net.L.reg8) + lane;
#pragma unroll
        for (int g4 = 0; g4 < NF / 4; ++g4) {
            const float4 v = p8[g4 * 64];
            frag[4 * g4] = v.x; frag[4 * g4 + 1] = v.y; frag[4 * g4 + 2] = v.z; frag[4 * g4 + 3] = v.w;
        }
    }
    float w[NL][2 * TPL * TPL], bias[NL][4 * TPL], wm[TPL], bm[4];
    {
        int fi = 0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int nk = (l == 0 ? 1 : TPL) + TPL;
#pragma unroll
            for (int k = 0; k < 2 * TPL; ++k)
#pragma unroll
                for (int t = 0; t < TPL; ++t) w[l][k * TPL + t] = k < nk ? frag[fi++] : 0.0f;
#pragma unroll
            for (int i = 0; i < 4 * TPL; ++i) bias[l][i] = frag[fi++];
        }
#pragma unroll
        for (int s2 = 0; s2 < TPL; ++s2) wm[s2] = frag[fi++];
#pragma unroll
        for (int r = 0; r < 4; ++r) bm[r] = frag[fi++];
    }
    const int act = net.d.mlp_act[0];
    if (has_slice) scene_update_slice_store(upd, dt, us);
    while (g < ngroups) {
        const float cx0 = x0, cx1 = x1, clim = lim_r, cvl = vl_r;
        const bool cvalid = valid;
        const int64_t crr = rr;
        float hl[NL][TPL], cl[NL][TPL];
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int s = 0; s < TPL; ++s) { hl[l][s] = hreg[l][s]; cl[l][s] = creg[l][s]; }
        int64_t gnext = g + nwaves;
        if (gnext < ngroups) load_group(gnext);
        float xin = gq == 0 ? cx0 : (gq == 1 ? cx1 : 0.0f);
        float prev[TPL] = {0.0f, 0.0f};
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            f32x4 acc[TPL];
#pragma unroll
            for (int t = 0; t < TPL; ++t) {
                acc[t][0] = bias[l][t * 4 + 0]; acc[t][1] = bias[l][t * 4 + 1];
                acc[t][2] = bias[l][t * 4 + 2]; acc[t][3] = bias[l][t * 4 + 3];
            }
            const int ks_in = l == 0 ? 1 : TPL;
            if (l == 0) {
#pragma unroll
                for (int t = 0; t < TPL; ++t) acc[t] = mfma4(w[l][t], xin, acc[t]);
            } else {
#pragma unroll
                for (int s = 0; s < TPL; ++s)
#pragma unroll
                    for (int t = 0; t < TPL; ++t) acc[t] = mfma4(w[l][s * TPL + t], prev[s], acc[t]);
            }
#pragma unroll
            for (int s = 0; s < TPL; ++s)
#pragma unroll
                for (int t = 0; t < TPL; ++t) acc[t] = mfma4(w[l][(ks_in + s) * TPL + t], hl[l][s], acc[t]);
            int64_t sbase = ((int64_t)l * rows + crr) * H + gq * TPL;
#pragma unroll
            for (int t = 0; t < TPL; ++t) {
                float ig = sigmoidf_(acc[t][0]);
                float fg = sigmoidf_(acc[t][1]);
                float gg = tanhf_(acc[t][2]);
                float og = sigmoidf_(acc[t][3]);
                float cn = fg * cl[l][t] + ig * gg;
                float hn = og * tanhf_(cn);
                prev[t] = hn;
                if (cvalid) { hs[sbase + t] = hn; cs[sbase + t] = cn; }
            }
        }
        f32x4 acc;
        acc[0] = bm[0]; acc[1] = bm[1]; acc[2] = bm[2]; acc[3] = bm[3];
#pragma unroll
        for (int s = 0; s < TPL; ++s) acc = mfma4(wm[s], prev[s], acc);
        float tau = act_fn(acc[0], act);
        if (cvalid && gq == 0) {
            // dc_clip_store with the limits already in registers
            float mx = tclamp(sat * (1.0f - cx1 / cvl), 0.0f, clim);
            float mn = tclamp(sat * (-1.0f - cx1 / cvl), -clim, 0.0f);
            computed[crr] = tau;
            applied[crr] = tclamp(tau, mn, mx);
        }
        g = gnext;
    }
    if (has_slice) {
        __syncthreads();          // every body thread of the slice has read the words the commit rewrites
        scene_update_slice_commit(upd, dt, us_e0, us);
    }
    MDP_STAMP(1);
}

__global__ void __launch_bounds__(256)
k_mlp(NetDev net, const float* __restrict__ packed, int64_t rows, int J, int Hlen,
      const float* __restrict__ q_des, int ld_q_des, const float* __restrict__ q, int ld_q,
      const float* __restrict__ qd, int ld_qd, float* __restrict__ pos_hist, float* __restrict__ vel_hist,
      const int32_t* __restrict__ input_idx, int n_idx, int order, float pos_scale, float vel_scale,
      float torque_scale, const float* __restrict__ eff_lim, const float* __restrict__ vel_lim, float sat,
      float* __restrict__ computed, float* __restrict__ applied) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < net.L.total; i += blockDim.x) lds[i] = packed[i];
    __syncthreads();
    int lane = threadIdx.x & 63;
    int gq = lane >> 4, col = lane & 15;
    int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    int64_t ngroups = (rows + 15) / 16;
    int F = 2 * n_idx;
    for (int64_t g = wave; g < ngroups; g += nwaves) {
        int64_t row = g * 16 + col;
        bool valid = row < rows;
        int64_t rr = valid ? row : rows - 1;
        int64_t e = rr / J;
        int j = (int)(rr - e * J);
        float perr = q_des[e * ld_q_des + j] - q[e * ld_q + j];  // actuator_net.py:153
        float vel = qd[e * ld_qd + j];
        // features 4*s + gq of the rolled histories (new[0] = current, new[h] = old[h-1])
        float in_first[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            int f = 4 * s + gq;
            float v = 0.0f;
            if (f < F) {
                bool first_half = f < n_idx;
                bool is_pos = (order == 0) ? first_half : !first_half;
                int tap = input_idx[first_half ? f : f - n_idx];
                const float* hist = is_pos ? pos_hist : vel_hist;
                float raw = tap == 0 ? (is_pos ? perr : vel) : hist[(e * Hlen + (tap - 1)) * J + j];
                v = raw * (is_pos ? pos_scale : vel_scale);
            }
            in_first[s] = v;
        }
        float tau = run_mlp(net, lds, lane, in_first) * torque_scale;
        if (valid && gq == 0) {
            dc_clip_store(tau, vel, row, eff_lim, vel_lim, sat, computed, applied);
            // roll by one (actuator_net.py:152-156); all taps above were consumed by the MFMA chain
            for (int h = Hlen - 1; h >= 1; --h) {
                pos_hist[(e * Hlen + h) * J + j] = pos_hist[(e * Hlen + h - 1) * J + j];
                vel_hist[(e * Hlen + h) * J + j] = vel_hist[(e * Hlen + h - 1) * J + j];
            }
            pos_hist[(e * Hlen) * J + j] = perr;
            vel_hist[(e * Hlen) * J + j] = vel;
        }
    }
}

__global__ void k_lstm_reset(int L, int H, int N, int J, float* __restrict__ hs, float* __restrict__ cs,
                             const uint8_t* __restrict__ mask) {
    int64_t per_layer = (int64_t)N * J * H;
    int64_t total = per_layer * L;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t within = i % per_layer;
        int e = (int)(within / ((int64_t)J * H));
        if (mask && !mask[e]) continue;
        hs[i] = 0.0f;
        cs[i] = 0.0f;
    }
}

}  // namespace

extern "C" int64_t mdp_net_packed_size(const mdp_net_desc_t* desc) {
    if (validate(desc) != MDP_OK) return -1;
    NetLayout L;
    return make_layout(*desc, L);
}

extern "C" int mdp_net_pack(const mdp_net_desc_t* desc, const float* const* weights, float* out) {
    int rc = validate(desc);
    if (rc != MDP_OK) return rc;
    MDP_REQUIRE(weights && out, "null pointer");
    const mdp_net_desc_t& d = *desc;
    NetLayout L;
    int total = make_layout(d, L);
    for (int i = 0; i < total; ++i) out[i] = 0.0f;
    int H = d.lstm_hidden, TPL = H > 0 ? H / 4 : 0;
    int wi = 0;
    for (int l = 0; l < d.lstm_layers; ++l) {
        int in_dim = l == 0 ? d.lstm_input : H;
        const float* w_ih = weights[wi++];
        const float* w_hh = weights[wi++];
        const float* b_ih = weights[wi++];
        const float* b_hh = weights[wi++];
        int ks_in = L.lstm_ksteps_in[l];
        for (int t = 0; t < TPL; ++t)
            for (int lane = 0; lane < 64; ++lane) {
                int m = lane & 15, kq = lane >> 4;
                int gq = m >> 2, r = m & 3;
                int grow = r * H + gq * TPL + t;  // PyTorch gate row (i,f,g,o blocks)
                // input k-steps
                for (int s = 0; s < ks_in; ++s) {
                    int in_idx = (l == 0) ? (4 * s + kq) : (kq * TPL + s);
                    float v = in_idx < in_dim ? w_ih[(int64_t)grow * in_dim + in_idx] : 0.0f;
                    out[L.lstm_w[l] + (s * TPL + t) * 64 + lane] = v;
                }
                for (int s = 0; s < TPL; ++s)
                    out[L.lstm_w[l] + ((ks_in + s) * TPL + t) * 64 + lane] = w_hh[(int64_t)grow * H + kq * TPL + s];
            }
        // bias fragments follow the D layout: lane (gq = lane>>4), register r
        for (int t = 0; t < TPL; ++t)
            for (int r = 0; r < 4; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    int gq = lane >> 4;
                    int grow = r * H + gq * TPL + t;
                    out[L.lstm_b[l] + (t * 4 + r) * 64 + lane] = b_ih[grow] + b_hh[grow];
                }
    }
    for (int l = 0; l < d.mlp_layers; ++l) {
        int in_dim = d.mlp_dims[l], out_dim = d.mlp_dims[l + 1];
        const float* w = weights[wi++];
        const float* b = weights[wi++];
        int ks = L.mlp_ksteps[l];
        for (int s = 0; s < ks; ++s)
            for (int t = 0; t < MLP_NT; ++t)
                for (int lane = 0; lane < 64; ++lane) {
                    int m = lane & 15, kq = lane >> 4;
                    int o = 16 * t + m;
                    int in_idx;
                    if (l == 0 && d.lstm_layers > 0) in_idx = kq * TPL + s;       // LSTM unit held by quad kq, slot s
                    else if (l == 0) in_idx = 4 * s + kq;                         // raw feature order
                    else in_idx = 16 * (s / 4) + 4 * kq + (s % 4);                // D layout of the previous layer
                    float v = (o < out_dim && in_idx < in_dim) ? w[(int64_t)o * in_dim + in_idx] : 0.0f;
                    out[L.mlp_w[l] + (s * MLP_NT + t) * 64 + lane] = v;
                }
        for (int t = 0; t < MLP_NT; ++t)
            for (int r = 0; r < 4; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    int o = 16 * t + 4 * (lane >> 4) + r;
                    out[L.mlp_b[l] + (t * 4 + r) * 64 + lane] = o < out_dim ? b[o] : 0.0f;
                }
    }
    if (L.reg8 >= 0) {      // the same numbers once more, four fragments of a lane side by side: one 16-byte load each
        int src[40];
        const int n = reg8_sources(d, L, src);
        for (int i = 0; i < n; ++i)
            for (int lane = 0; lane < 64; ++lane) out[L.reg8 + ((i / 4) * 64 + lane) * 4 + (i % 4)] = out[src[i] + lane];
    }
    return MDP_OK;
}

static unsigned net_grid(int64_t rows) {
    int64_t groups = (rows + 15) / 16;
    int64_t blocks = (groups + 3) / 4;  // 4 waves per block
    static const int64_t cap = [] { const char* e = std::getenv("MDP_NET_GRID_CAP"); return e ? (int64_t)std::atoll(e) : (int64_t)2048; }();
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

MDP_STAMP_SETTER(mdp_debug_stamps_actuator_nets)

static int lstm_launch(const mdp_joint_action_t* action, const mdp_scene_update_t* update, float dt,
                                                      const mdp_net_desc_t* desc, const float* packed_weights,
                                                      int num_envs, int num_joints, const float* q_des, int ld_q_des,
                                                      const float* joint_pos, int ld_pos, const float* joint_vel,
                                                      int ld_vel, float* hidden_state, float* cell_state,
                                                      const float* effort_limit, const float* velocity_limit,
                                                      float saturation_effort, float* computed_effort,
                                                      float* applied_effort, void* stream, const float* gate = nullptr) {
    int rc = validate(desc);
    if (rc != MDP_OK) return rc;
    MDP_REQUIRE(desc->lstm_layers >= 1, "not an LSTM network");
    MDP_REQUIRE(packed_weights && q_des && joint_pos && joint_vel && hidden_state && cell_state && effort_limit &&
                    velocity_limit && computed_effort && applied_effort,
                "null pointer");
    SceneUpdatePlan plan;
    if (update) {
        rc = scene_update_plan(update, &plan);
        if (rc != MDP_OK) return rc;
    } else {
        std::memset(&plan.d, 0, sizeof(plan.d));
    }
    int64_t rows = (int64_t)num_envs * num_joints;
    if (rows == 0) {
        if (update) return mdp_scene_update(update, dt, stream);
        return MDP_OK;
    }
    mdp_joint_action_t act;
    std::memset(&act, 0, sizeof(act));
    if (action) {
        MDP_REQUIRE(action->actions_in && action->action && action->processed, "joint action: null pointer");
        MDP_REQUIRE(action->ld_in >= num_joints, "Invalid action shape, expected: %d, received: %d.", num_joints,
                    action->ld_in);
        act = *action;
    }
    NetDev net;
    net.d = *desc;
    make_layout(*desc, net.L);
    size_t lds_bytes = (size_t)net.L.total * sizeof(float);
    MDP_REQUIRE(lds_bytes <= 150 * 1024, "network too large for LDS staging (%zu bytes)", lds_bytes);
    MDP_REQUIRE(((uintptr_t)packed_weights & 15) == 0, "packed_weights must be 16-byte aligned");
    unsigned nb_act = net_grid(rows);
    dim3 grid(nb_act + plan.carried_total), block(256);
    hipStream_t s = as_stream(stream);
    // register-resident kernel: the update rides inside the actuator blocks when a slice of envs per block fits one
    // round of their threads (MDP_UPD_INLINE=0: separate update blocks in front of the grid, as for the other kernels)
    static const bool inline_ok = [] { const char* e = std::getenv("MDP_UPD_INLINE"); return !e || e[0] != '0'; }();
    // Measured with the whole-step bench: at 16384 envs the launch goes 29.5 -> 25.8 us (step 208.6 -> 203.9 us); at
    // 8192 envs 15.4 -> 16.7 us and at 4096 envs the slice's registers (64 -> 86 VGPRs) cost more than the update blocks
    // did (76.8 -> 77.8 us/step when every launch used them), so smaller launches keep the update blocks and the
    // 64-register instantiation.  Beyond 2048 actuator blocks x 15 envs the slice no longer fits one round.
    int upd_mode = plan.carried_envs;
    dim3 grid8 = grid;
    static const int inline_min_envs = [] { const char* e = std::getenv("MDP_UPD_INLINE_MIN_ENVS"); return e ? std::atoi(e) : 12288; }();
    if (update && inline_ok && num_envs >= inline_min_envs) {
        const int n = scene_update_slice_envs(plan.d, nb_act, 256);
        if (n > 0) { upd_mode = -n; grid8 = dim3(nb_act); }
    }
#define LAUNCH(HH)                                                                                                 \
    do {                                                                                                           \
        if (lds_bytes > 64 * 1024)                                                                                 \
            MDP_CHECK_HIP(hipFuncSetAttribute((const void*)k_lstm<HH>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              (int)lds_bytes));                                                    \
        hipLaunchKernelGGL(k_lstm<HH>, grid, block, lds_bytes, s, plan.d, dt, nb_act, plan.nb_root, plan.nb_joint, \
                           plan.nb_contact, plan.carried_envs, act, net, packed_weights, rows, num_joints, q_des, ld_q_des, joint_pos,     \
                           ld_pos, joint_vel, ld_vel, hidden_state, cell_state, effort_limit, velocity_limit,      \
                           saturation_effort, computed_effort, applied_effort, gate);                              \
    } while (0)
    // register-resident weights for the ANYdrive shape (MDP_LSTM_GENERIC=1 forces the LDS kernel: parity tests)
    static const bool force_generic = [] { const char* e = std::getenv("MDP_LSTM_GENERIC"); return e && e[0] == '1'; }();
    if (!force_generic && net.L.reg8 >= 0 && rows < (1ll << 31)) {
        RaycasterUpdateArgs no_cast;
        std::memset(&no_cast, 0, sizeof(no_cast));
#define L8(NLAYERS)                                                                                                 \
    do {                                                                                                            \
        if (upd_mode < 0)                                                                                           \
            hipLaunchKernelGGL((k_lstm8_reg<NLAYERS, true, false>), grid8, block, 0, s, no_cast, 0u, plan.d, dt, nb_act, plan.nb_root,  \
                               plan.nb_joint, plan.nb_contact, upd_mode, act, net, packed_weights, rows, num_joints, \
                               q_des, ld_q_des, joint_pos, ld_pos, joint_vel, ld_vel, hidden_state, cell_state,     \
                               effort_limit, velocity_limit, saturation_effort, computed_effort, applied_effort,    \
                               gate);                                                                               \
        else                                                                                                        \
            hipLaunchKernelGGL((k_lstm8_reg<NLAYERS, false, false>), grid8, block, 0, s, no_cast, 0u, plan.d, dt, nb_act, plan.nb_root, \
                               plan.nb_joint, plan.nb_contact, upd_mode, act, net, packed_weights, rows, num_joints, \
                               q_des, ld_q_des, joint_pos, ld_pos, joint_vel, ld_vel, hidden_state, cell_state,     \
                               effort_limit, velocity_limit, saturation_effort, computed_effort, applied_effort,    \
                               gate);                                                                               \
    } while (0)
        if (desc->lstm_layers == 1)
            L8(1);
        else
            L8(2);
#undef L8
        MDP_LAUNCH_CHECK();
        return MDP_OK;
    }
    switch (desc->lstm_hidden) {
        case 8: LAUNCH(8); break;
        case 16: LAUNCH(16); break;
        case 32: LAUNCH(32); break;
        default: return fail(MDP_ERR_UNSUPPORTED, "lstm hidden %d", desc->lstm_hidden);
    }
#undef LAUNCH
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_actuator_lstm_compute_after_update(const mdp_scene_update_t* update, float dt,
                                                      const mdp_net_desc_t* desc, const float* packed_weights,
                                                      int num_envs, int num_joints, const float* q_des, int ld_q_des,
                                                      const float* joint_pos, int ld_pos, const float* joint_vel,
                                                      int ld_vel, float* hidden_state, float* cell_state,
                                                      const float* effort_limit, const float* velocity_limit,
                                                      float saturation_effort, float* computed_effort,
                                                      float* applied_effort, void* stream) {
    return lstm_launch(nullptr, update, dt, desc, packed_weights, num_envs, num_joints, q_des, ld_q_des, joint_pos, ld_pos,
                       joint_vel, ld_vel, hidden_state, cell_state, effort_limit, velocity_limit, saturation_effort,
                       computed_effort, applied_effort, stream);
}

extern "C" int mdp_actuator_lstm_compute_with_action(const mdp_joint_action_t* action, const mdp_net_desc_t* desc,
                                                     const float* packed_weights, int num_envs, int num_joints,
                                                     const float* joint_pos, int ld_pos, const float* joint_vel,
                                                     int ld_vel, float* hidden_state, float* cell_state,
                                                     const float* effort_limit, const float* velocity_limit,
                                                     float saturation_effort, float* computed_effort,
                                                     float* applied_effort, void* stream) {
    MDP_REQUIRE(action, "null joint action");
    return lstm_launch(action, nullptr, 0.0f, desc, packed_weights, num_envs, num_joints, action->processed, num_joints,
                       joint_pos, ld_pos, joint_vel, ld_vel, hidden_state, cell_state, effort_limit, velocity_limit,
                       saturation_effort, computed_effort, applied_effort, stream);
}

extern "C" int mdp_actuator_lstm_compute(const mdp_net_desc_t* desc, const float* packed_weights, int num_envs,
                                         int num_joints, const float* q_des, int ld_q_des, const float* joint_pos,
                                         int ld_pos, const float* joint_vel, int ld_vel, float* hidden_state,
                                         float* cell_state, const float* effort_limit, const float* velocity_limit,
                                         float saturation_effort, float* computed_effort, float* applied_effort,
                                         void* stream) {
    return mdp_actuator_lstm_compute_after_update(nullptr, 0.0f, desc, packed_weights, num_envs, num_joints, q_des,
                                                  ld_q_des, joint_pos, ld_pos, joint_vel, ld_vel, hidden_state,
                                                  cell_state, effort_limit, velocity_limit, saturation_effort,
                                                  computed_effort, applied_effort, stream);
}

extern "C" int mdp_actuator_lstm_compute_gated(const float* gate, const mdp_net_desc_t* desc, const float* packed_weights,
                                               int num_envs, int num_joints, const float* q_des, int ld_q_des,
                                               const float* joint_pos, int ld_pos, const float* joint_vel, int ld_vel,
                                               float* hidden_state, float* cell_state, const float* effort_limit,
                                               const float* velocity_limit, float saturation_effort,
                                               float* computed_effort, float* applied_effort, void* stream) {
    MDP_REQUIRE(gate, "null gate");
    return lstm_launch(nullptr, nullptr, 0.0f, desc, packed_weights, num_envs, num_joints, q_des, ld_q_des, joint_pos, ld_pos,
                       joint_vel, ld_vel, hidden_state, cell_state, effort_limit, velocity_limit, saturation_effort,
                       computed_effort, applied_effort, stream, gate);
}

extern "C" int mdp_actuator_lstm_compute_gated_with_recast(const float* gate, const mdp_net_desc_t* desc,
                                                           const float* packed_weights, int num_envs, int num_joints,
                                                           const float* q_des, int ld_q_des, const float* joint_pos,
                                                           int ld_pos, const float* joint_vel, int ld_vel,
                                                           float* hidden_state, float* cell_state,
                                                           const float* effort_limit, const float* velocity_limit,
                                                           float saturation_effort, float* computed_effort,
                                                           float* applied_effort,
                                                           const mdp_raycaster_update_args_t* cast,
                                                           const uint8_t* reset_mask, void* stream) {
    MDP_REQUIRE(gate, "null gate");
    int rc = validate(desc);
    if (rc != MDP_OK) return rc;
    MDP_REQUIRE(packed_weights && q_des && joint_pos && joint_vel && hidden_state && cell_state && effort_limit &&
                    velocity_limit && computed_effort && applied_effort, "null pointer");
    const int64_t rows = (int64_t)num_envs * num_joints;
    if (!(desc->lstm_layers >= 1 && desc->lstm_hidden == 8 && desc->lstm_layers <= 2 && desc->mlp_layers == 1 &&
          rows > 0 && rows < (1ll << 31)))
        return fail(MDP_ERR_UNSUPPORTED, "the fused reset tail covers the register-resident LSTM (hidden 8, <= 2 layers, linear decoder)");
    MDP_REQUIRE(((uintptr_t)packed_weights & 15) == 0, "packed_weights must be 16-byte aligned");
    RaycasterUpdateArgs ra;
    rc = raycaster_flagged_args(num_envs, cast, reset_mask, &ra);
    if (rc != MDP_OK) return rc;
    MDP_REQUIRE(cast->num_rays <= 256, "the fused reset tail casts patterns of up to 256 rays per env block round");
    NetDev net;
    net.d = *desc;
    make_layout(*desc, net.L);
    const unsigned nb_act = net_grid(rows);
    const unsigned nb_cast = (unsigned)(((int64_t)num_envs + RCF_ENVS - 1) / RCF_ENVS);
    hipStream_t s = as_stream(stream);
    mdp_scene_update_t no_upd;                   // (no carried update, no action in this launch)
    mdp_joint_action_t no_act;
    std::memset(&no_upd, 0, sizeof(no_upd));
    std::memset(&no_act, 0, sizeof(no_act));
    if (desc->lstm_layers == 1)
        hipLaunchKernelGGL((k_lstm8_reg<1, false, true>), dim3(nb_cast + nb_act), dim3(256), 0, s, ra, nb_cast, no_upd, 0.0f,
                           nb_act, 0u, 0u, 0u, 0, no_act, net, packed_weights, rows, num_joints, q_des, ld_q_des, joint_pos,
                           ld_pos, joint_vel, ld_vel, hidden_state, cell_state, effort_limit, velocity_limit,
                           saturation_effort, computed_effort, applied_effort, gate);
    else
        hipLaunchKernelGGL((k_lstm8_reg<2, false, true>), dim3(nb_cast + nb_act), dim3(256), 0, s, ra, nb_cast, no_upd, 0.0f,
                           nb_act, 0u, 0u, 0u, 0, no_act, net, packed_weights, rows, num_joints, q_des, ld_q_des, joint_pos,
                           ld_pos, joint_vel, ld_vel, hidden_state, cell_state, effort_limit, velocity_limit,
                           saturation_effort, computed_effort, applied_effort, gate);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_actuator_lstm_reset(int layers, int hidden, int num_envs, int num_joints, float* hidden_state,
                                       float* cell_state, const uint8_t* env_mask, void* stream) {
    MDP_REQUIRE(hidden_state && cell_state, "null pointer");
    int64_t total = (int64_t)layers * num_envs * num_joints * hidden;
    if (total == 0) return MDP_OK;
    hipLaunchKernelGGL(k_lstm_reset, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), layers, hidden,
                       num_envs, num_joints, hidden_state, cell_state, env_mask);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}

extern "C" int mdp_actuator_mlp_compute(const mdp_net_desc_t* desc, const float* packed_weights, int num_envs,
                                        int num_joints, int history_len, const float* q_des, int ld_q_des,
                                        const float* joint_pos, int ld_pos, const float* joint_vel, int ld_vel,
                                        float* pos_err_history, float* vel_history, const int32_t* input_idx,
                                        int n_idx, int input_order, float pos_scale, float vel_scale,
                                        float torque_scale, const float* effort_limit, const float* velocity_limit,
                                        float saturation_effort, float* computed_effort, float* applied_effort,
                                        void* stream) {
    int rc = validate(desc);
    if (rc != MDP_OK) return rc;
    MDP_REQUIRE(desc->lstm_layers == 0, "not a pure MLP network");
    MDP_REQUIRE(n_idx >= 1 && 2 * n_idx == desc->mlp_dims[0], "network input width %d != 2*len(input_idx)=%d",
                desc->mlp_dims[0], 2 * n_idx);
    MDP_REQUIRE(input_order == 0 || input_order == 1,
                "Invalid input order for MLP actuator net. Must be 'pos_vel' or 'vel_pos'.");
    MDP_REQUIRE(packed_weights && q_des && joint_pos && joint_vel && pos_err_history && vel_history && input_idx &&
                    effort_limit && velocity_limit && computed_effort && applied_effort,
                "null pointer");
    int64_t rows = (int64_t)num_envs * num_joints;
    if (rows == 0) return MDP_OK;
    NetDev net;
    net.d = *desc;
    make_layout(*desc, net.L);
    size_t lds_bytes = (size_t)net.L.total * sizeof(float);
    hipLaunchKernelGGL(k_mlp, dim3(net_grid(rows)), dim3(256), lds_bytes, as_stream(stream), net, packed_weights,
                       rows, num_joints, history_len, q_des, ld_q_des, joint_pos, ld_pos, joint_vel, ld_vel,
                       pos_err_history, vel_history, input_idx, n_idx, input_order, pos_scale, vel_scale,
                       torque_scale, effort_limit, velocity_limit, saturation_effort, computed_effort,
                       applied_effort);
    MDP_LAUNCH_CHECK();
    return MDP_OK;
}
