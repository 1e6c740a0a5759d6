// LCMHelpers.hpp -- the on-the-wire contract between the MPC loop and the robot-side trajectory runner, without
// the LCM transport: the message types of lcmtypes/lcmt_trajectory_{f,d}.lcm, lcmt_solver_params.lcm, lcmt_cost_params.lcm with LCM's standard
// binary encoding, how the MPC loop fills a trajectory message (LCM_MPCLoop_Handler::handleStatus, DDPHelpers/LCMHelpers.cuh:239-262) and what the
// trajectory runner does with it (LCM_TrajRunner, :98-153).  Sockets, channels, multicast and the Drake status/command types stay outside.
//
// Encoding (LCM wire format of a message body): 8-byte fingerprint, then the members in declaration order, every scalar big-endian, arrays as
// consecutive scalars.  fingerprint = rotl1(base) with base = lcm-gen's hash over (member name, primitive type name, dimensions) starting from
// 0x12345678 -- restated from the LCM generator's published algorithm.  PINNED: the four fingerprints equal rotl1 of the base hashes in the
// reference's own lcm-gen output (lcmtypes/drake/lcmt_trajectory_{f,d}.hpp:212, lcmtypes/kuka/lcmt_cost_params.hpp:304, lcmt_solver_params.hpp:178;
// fixture tests/golden/lcm_hashes.json, test tests/test_wire_format.py).
//
// A reference quirk that IS the contract: the MPC loop stores BYTE counts in x_size / u_size / KT_size (ld * TRAJ_RUNNER_TIME_STEPS * sizeof(T),
// :241-246) and sizes the arrays with them, so a message carries sizeof(T) times more elements than the trajectory has (the tail is zero); the
// runner copies `size` bytes back (:121-125).  pack / unpack below do exactly that.
#ifndef PDDP_HOSTAPI_LCMHELPERS_HPP
#define PDDP_HOSTAPI_LCMHELPERS_HPP

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace pddp_wire {

// ---- lcm-gen's structure hash
inline int64_t hash_update(int64_t v, char c) { return static_cast<int64_t>((static_cast<uint64_t>(v) << 8) ^ static_cast<uint64_t>(v >> 55)) + c; }
inline int64_t hash_string_update(int64_t v, const char* s) {
    v = hash_update(v, static_cast<char>(std::strlen(s)));
    for (; *s != 0; s++) v = hash_update(v, *s);
    return v;
}
struct Member { const char* name; const char* type; const char* var_dim; };   // var_dim: name of the size member of a 1-D variable array, or nullptr
inline int64_t fingerprint(const Member* m, int count) {
    int64_t v = 0x12345678;
    for (int i = 0; i < count; i++) {
        v = hash_string_update(v, m[i].name);
        v = hash_string_update(v, m[i].type);               // every member of these types is a primitive
        v = hash_update(v, m[i].var_dim ? 1 : 0);           // number of dimensions
        if (m[i].var_dim) { v = hash_update(v, 1 /* LCM_VAR */); v = hash_string_update(v, m[i].var_dim); }
    }
    const uint64_t h = static_cast<uint64_t>(v);
    return static_cast<int64_t>((h << 1) + ((h >> 63) & 1));
}

// ---- big-endian scalars
inline void put_u32(std::vector<uint8_t>& b, uint32_t v) { for (int s = 24; s >= 0; s -= 8) b.push_back(static_cast<uint8_t>(v >> s)); }
inline void put_u64(std::vector<uint8_t>& b, uint64_t v) { for (int s = 56; s >= 0; s -= 8) b.push_back(static_cast<uint8_t>(v >> s)); }
inline void put_f32(std::vector<uint8_t>& b, float f) { uint32_t v; std::memcpy(&v, &f, 4); put_u32(b, v); }
inline void put_f64(std::vector<uint8_t>& b, double f) { uint64_t v; std::memcpy(&v, &f, 8); put_u64(b, v); }
struct Reader {
    const uint8_t* p; size_t n, pos; bool ok;
    Reader(const uint8_t* d, size_t len) : p(d), n(len), pos(0), ok(true) {}
    uint32_t u32() { if (pos + 4 > n) { ok = false; return 0; } uint32_t v = 0; for (int i = 0; i < 4; i++) v = (v << 8) | p[pos++]; return v; }
    uint64_t u64() { if (pos + 8 > n) { ok = false; return 0; } uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[pos++]; return v; }
    float f32() { const uint32_t v = u32(); float f; std::memcpy(&f, &v, 4); return f; }
    double f64() { const uint64_t v = u64(); double f; std::memcpy(&f, &v, 8); return f; }
};
template <typename S> inline void put_real(std::vector<uint8_t>& b, S v);
template <> inline void put_real<float>(std::vector<uint8_t>& b, float v) { put_f32(b, v); }
template <> inline void put_real<double>(std::vector<uint8_t>& b, double v) { put_f64(b, v); }
template <typename S> inline S get_real(Reader& r);
template <> inline float get_real<float>(Reader& r) { return r.f32(); }
template <> inline double get_real<double>(Reader& r) { return r.f64(); }

// ---- lcmtypes/lcmt_trajectory_f.lcm / lcmt_trajectory_d.lcm
template <typename S>
struct lcmt_trajectory {
    int64_t utime;
    int32_t x_size, u_size, KT_size;
    std::vector<S> x, u, KT;
    static int64_t getHash() {
        const char* t = sizeof(S) == 4 ? "float" : "double";
        const Member m[7] = {{"utime", "int64_t", nullptr}, {"x_size", "int32_t", nullptr}, {"u_size", "int32_t", nullptr}, {"KT_size", "int32_t", nullptr},
                             {"x", t, "x_size"}, {"u", t, "u_size"}, {"KT", t, "KT_size"}};
        return fingerprint(m, 7);
    }
    std::vector<uint8_t> encode() const {
        std::vector<uint8_t> b;
        put_u64(b, static_cast<uint64_t>(getHash())); put_u64(b, static_cast<uint64_t>(utime));
        put_u32(b, static_cast<uint32_t>(x_size)); put_u32(b, static_cast<uint32_t>(u_size)); put_u32(b, static_cast<uint32_t>(KT_size));
        for (int32_t i = 0; i < x_size; i++) put_real<S>(b, x[i]);
        for (int32_t i = 0; i < u_size; i++) put_real<S>(b, u[i]);
        for (int32_t i = 0; i < KT_size; i++) put_real<S>(b, KT[i]);
        return b;
    }
    bool decode(const uint8_t* data, size_t len) {
        Reader r(data, len);
        if (static_cast<int64_t>(r.u64()) != getHash()) return false;
        utime = static_cast<int64_t>(r.u64());
        x_size = static_cast<int32_t>(r.u32()); u_size = static_cast<int32_t>(r.u32()); KT_size = static_cast<int32_t>(r.u32());
        if (!r.ok || x_size < 0 || u_size < 0 || KT_size < 0) return false;
        if (r.pos + (static_cast<size_t>(x_size) + u_size + KT_size) * sizeof(S) > len) return false;
        x.resize(x_size); u.resize(u_size); KT.resize(KT_size);
        for (auto& v : x) v = get_real<S>(r);
        for (auto& v : u) v = get_real<S>(r);
        for (auto& v : KT) v = get_real<S>(r);
        return r.ok;
    }
};
typedef lcmt_trajectory<float> lcmt_trajectory_f;
typedef lcmt_trajectory<double> lcmt_trajectory_d;

// ---- lcmtypes/lcmt_solver_params.lcm
struct lcmt_solver_params {
    int64_t utime; int32_t iterLimit, timeLimit, clearVars, useCostShift;
    static int64_t getHash() {
        const Member m[5] = {{"utime", "int64_t", nullptr}, {"iterLimit", "int32_t", nullptr}, {"timeLimit", "int32_t", nullptr}, {"clearVars", "int32_t", nullptr},
                             {"useCostShift", "int32_t", nullptr}};
        return fingerprint(m, 5);
    }
    std::vector<uint8_t> encode() const {
        std::vector<uint8_t> b; put_u64(b, static_cast<uint64_t>(getHash())); put_u64(b, static_cast<uint64_t>(utime));
        put_u32(b, iterLimit); put_u32(b, timeLimit); put_u32(b, clearVars); put_u32(b, useCostShift); return b;
    }
    bool decode(const uint8_t* d, size_t len) {
        Reader r(d, len); if (static_cast<int64_t>(r.u64()) != getHash()) return false;
        utime = static_cast<int64_t>(r.u64()); iterLimit = r.u32(); timeLimit = r.u32(); clearVars = r.u32(); useCostShift = r.u32(); return r.ok;
    }
};

// ---- lcmtypes/lcmt_cost_params.lcm (member order of the .lcm file)
struct lcmt_cost_params {
    int64_t utime;
    float q_ee1, q_ee2, qf_ee1, qf_ee2, q_eev1, q_eev2, qf_eev1, qf_eev2, q_xdee, qf_xdee, q_xee, qf_xee, r_ee, q1, q2, qf1, qf2, r;
    static const char* const* names() {
        static const char* const n[18] = {"q_ee1", "q_ee2", "qf_ee1", "qf_ee2", "q_eev1", "q_eev2", "qf_eev1", "qf_eev2", "q_xdee", "qf_xdee", "q_xee", "qf_xee", "r_ee",
                                          "q1", "q2", "qf1", "qf2", "r"};
        return n;
    }
    float* fields() { return &q_ee1; }
    const float* fields() const { return &q_ee1; }
    static int64_t getHash() {
        Member m[19]; m[0] = {"utime", "int64_t", nullptr};
        for (int i = 0; i < 18; i++) m[1 + i] = {names()[i], "float", nullptr};
        return fingerprint(m, 19);
    }
    std::vector<uint8_t> encode() const {
        std::vector<uint8_t> b; put_u64(b, static_cast<uint64_t>(getHash())); put_u64(b, static_cast<uint64_t>(utime));
        for (int i = 0; i < 18; i++) put_f32(b, fields()[i]);
        return b;
    }
    bool decode(const uint8_t* d, size_t len) {
        Reader r(d, len); if (static_cast<int64_t>(r.u64()) != getHash()) return false;
        utime = static_cast<int64_t>(r.u64());
        for (int i = 0; i < 18; i++) fields()[i] = r.f32();
        return r.ok;
    }
};

// ---- what the MPC loop publishes after a solve (LCM_MPCLoop_Handler::handleStatus, LCMHelpers.cuh:239-262): sizes are BYTE counts, sic
template <typename S>
lcmt_trajectory<S> packTrajectory(const S* x, const S* u, const S* KT, int ld_x, int ld_u, int ld_KT, int dim_KT_c, int time_steps, int64_t t0_plant,
                                  bool use_feedback_in_traj_runner = true) {
    lcmt_trajectory<S> m;
    m.utime = t0_plant;
    const int stepsSize = time_steps * static_cast<int>(sizeof(S));
    m.u_size = ld_u * stepsSize; m.u.assign(m.u_size, S(0)); std::memcpy(m.u.data(), u, m.u_size);
    if (use_feedback_in_traj_runner) {
        m.x_size = ld_x * stepsSize; m.x.assign(m.x_size, S(0)); std::memcpy(m.x.data(), x, m.x_size);
        m.KT_size = ld_KT * dim_KT_c * stepsSize; m.KT.assign(m.KT_size, S(0)); std::memcpy(m.KT.data(), KT, m.KT_size);
    } else { m.x_size = 0; m.KT_size = 0; }
    return m;
}
// ---- what the trajectory runner keeps of a message (LCM_TrajRunner::newTrajCallback_{f,d}, :121-125): `size` BYTES of each array
template <typename S>
void unpackTrajectory(const lcmt_trajectory<S>& m, S* x, S* u, S* KT, int64_t* t0) {
    *t0 = m.utime;
    std::memcpy(u, m.u.data(), m.u_size); std::memcpy(x, m.x.data(), m.x_size); std::memcpy(KT, m.KT.data(), m.KT_size);
}

}  // namespace pddp_wire

#ifdef PDDP_HOSTAPI_MPCHELPERS_HPP
// ---- the robot-side trajectory runner (LCM_TrajRunner, LCMHelpers.cuh:98-153) without the transport: keeps the latest trajectory message and
// turns a measured state into a torque command with getHardwareControls (zero-order hold on u and K, first-order hold on x, output smoothing)
template <typename T>
class TrajRunner {
  public:
    std::vector<T> x, u, KT;
    int ld_x, ld_u, ld_KT;
    int64_t t0;
    bool ready, PDMode;
    std::vector<double> q_prev, u_prev;
    double alpha;                                        // smoothing: 1 = all old, 0 = all new
    TrajRunner(int _ld_x, int _ld_u, int _ld_KT, double a = 0.5, bool PD = false)
        : x((size_t)_ld_x * NUM_TIME_STEPS), u((size_t)_ld_x * NUM_TIME_STEPS), KT((size_t)_ld_KT * DIM_KT_c * NUM_TIME_STEPS), ld_x(_ld_x), ld_u(_ld_u), ld_KT(_ld_KT),
          t0(0), ready(false), PDMode(PD), q_prev(NUM_POS, 0.0), u_prev(CONTROL_SIZE, 0.0), alpha(a) {}
    void newTrajCallback(const pddp_wire::lcmt_trajectory<T>& msg) { pddp_wire::unpackTrajectory<T>(msg, x.data(), u.data(), KT.data(), &t0); ready = true; }
    // returns 0 and fills joint_position[NUM_POS] / joint_torque[NUM_POS]; 1 when not ready or asked beyond the trajectory (nothing is published then)
    int statusCallback(const double* joint_position_measured, const double* joint_velocity_estimated, int64_t utime, double* joint_position, double* joint_torque) {
        if (!ready) return 1;
        const int err = getHardwareControls<T>(joint_position, joint_torque, x.data(), u.data(), KT.data(), static_cast<double>(t0), joint_position_measured,
                                               joint_velocity_estimated, static_cast<double>(utime), ld_x, ld_u, ld_KT, q_prev.data(), u_prev.data(), alpha);
        if (PDMode) for (int i = 0; i < NUM_POS; i++) joint_position[i] = 0.5 * (static_cast<double>(x[(size_t)(TRAJ_RUNNER_TIME_STEPS - 1) * ld_x + i]) + joint_position_measured[i]);
        if (err) std::printf("[!]CRITICAL ERROR: Asked to execute beyond bounds of current traj.\n");
        return err;
    }
};
// the message the MPC loop publishes for the trajectory runner after runiLQR_MPC_GPU (LCM_MPCLoop_Handler::handleStatus, :239-262)
template <typename T>
pddp_wire::lcmt_trajectory<T> trajectoryMessage(const trajVars<T>* tvars, const matDimms* dimms) {
    return pddp_wire::packTrajectory<T>(tvars->x, tvars->u, tvars->KT, dimms->ld_x, dimms->ld_u, dimms->ld_KT, DIM_KT_c, TRAJ_RUNNER_TIME_STEPS, tvars->t0_plant,
                                        USE_FEEDBACK_IN_TRAJ_RUNNER != 0);
}
// lcmt_cost_params / lcmt_solver_params into the solver's records (handleCostParams / handleSolverParams, :203-214)
template <typename T>
void applyCostParams(costParams<T>* cst, const pddp_wire::lcmt_cost_params& m) {
    cst->Q_EE1 = m.q_ee1; cst->Q_EE2 = m.q_ee2; cst->QF_EE1 = m.qf_ee1; cst->QF_EE2 = m.qf_ee2; cst->Q_EEV1 = m.q_eev1; cst->Q_EEV2 = m.q_eev2; cst->QF_EEV1 = m.qf_eev1;
    cst->QF_EEV2 = m.qf_eev2; cst->Q_xdEE = m.q_xdee; cst->QF_xdEE = m.qf_xdee; cst->Q_xEE = m.q_xee; cst->QF_xEE = m.qf_xee; cst->R_EE = m.r_ee;
    cst->Q1 = m.q1; cst->Q2 = m.q2; cst->QF1 = m.qf1; cst->QF2 = m.qf2; cst->R = m.r;
}
#endif

#endif
