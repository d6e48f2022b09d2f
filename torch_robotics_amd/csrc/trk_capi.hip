// trk_capi.hip -- C ABI of libtrk.so (include/trk.h): argument checking, model / cost-model
// handles (host tables -> device tables, copied once), kernel launches.  No torch types here.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>
#include "trk_launch.h"
#include "trk_spec_common.h"

namespace {
thread_local std::string g_err;
unsigned long long* g_stamps = nullptr;   // profiling hook, see trk_debug_set_stamp_buffer
// Which kernel family served this thread's latest rollout call (trk_last_dispatch), and the strict mode in which a rollout entry
// point refuses the table-driven kernels for a model that HAS generated units (trk_set_strict_specialized / TRK_STRICT_SPECIALIZED=1):
// the table-driven fused rollout is 10 - 30 x slower, and a planner that silently lands on it has a configuration problem.
thread_local int g_last_dispatch = TRK_DISPATCH_NONE;
int g_strict = -1;                        // -1: not decided yet (the environment is read once)
bool strict_specialized() {
    if (g_strict < 0) { const char* e = std::getenv("TRK_STRICT_SPECIALIZED"); g_strict = (e && std::atoi(e) != 0) ? 1 : 0; }
    return g_strict == 1;
}

int fail(int code, const std::string& msg) { g_err = msg; return code; }
int hip_fail(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return TRK_ERR_HIP;
}
#define TRK_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_, #call); } while (0)

bool g_init_done = false;
int ensure_init() {
    if (g_init_done) return TRK_OK;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(TRK_ERR_NO_DEVICE, "no HIP device visible (libtrk.so needs an MI355X / gfx950 GPU)");
    }
    if (trk_kernels_init() != 0) return fail(TRK_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    g_init_done = true;
    return TRK_OK;
}

uint64_t fnv1a(uint64_t h, const void* data, size_t n) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}
// must hash the same bytes, in the same order, as torch_robotics_amd/codegen.py: model_hash
uint64_t model_hash(const TrkKinModelDesc* d) {
    const int L = d->n_links;
    uint64_t h = 0xcbf29ce484222325ull;
    const int32_t hd[2] = {d->n_links, d->n_dofs};
    h = fnv1a(h, hd, sizeof(hd));
    h = fnv1a(h, d->parent, 4 * L); h = fnv1a(h, d->joint_type, 4 * L); h = fnv1a(h, d->dof_idx, 4 * L);
    h = fnv1a(h, d->R_fixed, 36 * L); h = fnv1a(h, d->trans, 12 * L); h = fnv1a(h, d->axis, 12 * L);
    h = fnv1a(h, d->rot_axis, 4 * L); h = fnv1a(h, d->rot_sign, 4 * L); h = fnv1a(h, d->clamp, 4 * L);
    h = fnv1a(h, d->lower, 4 * L); h = fnv1a(h, d->upper, 4 * L); h = fnv1a(h, d->order, 4 * L);
    return h;
}
std::vector<const SpecEntry*>& spec_registry() { static std::vector<const SpecEntry*> r; return r; }
}  // namespace

// Registry of live handles: a dispatcher op (csrc/trk_torch_ops.cpp) receives handles as plain integers, and a stale integer must be
// an error, not a segfault -- trk_handle_kind answers for any pointer value without dereferencing it.
namespace {
std::mutex g_live_mu;
std::unordered_map<const void*, int> g_live;             // pointer -> 1 model, 2 cost model, 3 point set
void live_add(const void* p, int kind) { std::lock_guard<std::mutex> l(g_live_mu); g_live[p] = kind; }
void live_del(const void* p) { std::lock_guard<std::mutex> l(g_live_mu); g_live.erase(p); }
}  // namespace
int trk_handle_kind(const void* handle) {
    std::lock_guard<std::mutex> l(g_live_mu);
    auto it = g_live.find(handle);
    return it == g_live.end() ? 0 : it->second;
}

// error reporting / one-time initialisation for the library's other translation units (trk_exchange.hip)
int trk_fail(int code, const char* msg) { return fail(code, msg); }
int trk_hip_fail(int hip_error, const char* what) { return hip_fail((hipError_t)hip_error, what); }
int trk_ensure_init(void) { return ensure_init(); }

int trk_spec_register(const SpecEntry* e) {
    // the first four fields are the layout stamp in every version of SpecEntry; nothing else is read before they match
    if (!e || e->spec_abi_version != TRK_SPEC_ABI_VERSION || e->sizeof_args != sizeof(SpecArgs) + sizeof(IkArgs) + sizeof(IkGnArgs) ||
        e->sizeof_entry != sizeof(SpecEntry) || e->sizeof_cost_hdr != sizeof(DevCostHdr)) {
        fprintf(stderr, "libtrk: refusing a generated unit compiled against another SpecArgs/SpecEntry layout "
                        "(stale JIT cache?) -- it will not be dispatched\n");
        return TRK_ERR_INVALID_ARG;
    }
    spec_registry().push_back(e);
    return 0;
}
int trk_spec_count(void) { return (int)spec_registry().size(); }
int trk_spec_layout_stamp(int64_t* out) {
    if (!out) return TRK_ERR_INVALID_ARG;
    out[0] = TRK_SPEC_ABI_VERSION; out[1] = (int64_t)(sizeof(SpecArgs) + sizeof(IkArgs) + sizeof(IkGnArgs)); out[2] = (int64_t)sizeof(DevCostHdr);
    return 0;
}
const SpecEntry* trk_spec_find(uint64_t h, int n_links, int n_dofs) {
    for (const SpecEntry* e : spec_registry())
        if (e->n_points == 0 && e->model_hash == h && e->n_links == n_links && e->n_dofs == n_dofs) return e;
    return nullptr;
}
const SpecEntry* trk_spec_find_points(uint64_t h, uint64_t points_hash, int n_points) {
    for (const SpecEntry* e : spec_registry())
        if (e->n_points == n_points && n_points > 0 && e->model_hash == h && e->points_hash == points_hash) return e;
    return nullptr;
}

// ------------------------------------------------------------------------------------------------------------------------------
// A generated unit loaded as a CODE OBJECT (jit.py's hipRTC fall-back: the unit's device half compiled in-process when no hipcc is
// around).  The unit's host half -- its launchers -- is played by the generic functions below: same kernel choice by (I/O mode,
// compile-time switches, base pose), same grids, looked up in a table of hipFunction_t filled at registration.
// ------------------------------------------------------------------------------------------------------------------------------
namespace {
struct ModuleUnit {
    hipModule_t mod = nullptr;
    std::string ident;
    std::vector<int32_t> obj, pairs, vsrc;
    std::vector<float> vw;
    bool chunked = false, fast_switch = false, jac_direct = false;
    int D = 0;
    hipFunction_t rollout[3][2][2][2] = {};     // [io][POS (chunked units, else 0)][FAST or BOX][base identity]
    hipFunction_t gp[3][2][2] = {};             // [io][FAST or BOX][base identity]
    hipFunction_t posbwd[2] = {}, coll[2] = {}, fkh[2] = {}, fkhbwd[2] = {}, fk1[2] = {}, ik[2] = {}, ikgn[2] = {}, jac[2] = {};
    hipFunction_t fields = nullptr, collf = nullptr;
    hipFunction_t prollout[2][2] = {};          // attached-point units: [FAST][base identity]
    SpecEntry entry{};
};
std::vector<ModuleUnit*>& module_units() { static std::vector<ModuleUnit*> v; return v; }

thread_local hipError_t g_module_launch_error = hipSuccess;     // the latest failure of a code-object launch ON THIS THREAD (like g_err): picked up by last_launch_error()
hipError_t last_launch_error() {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = g_module_launch_error;
    g_module_launch_error = hipSuccess;
    return e;
}
template <class Args>
void mod_launch(hipFunction_t f, unsigned grid, unsigned block, size_t lds, const Args& a, hipStream_t st) {
    Args copy = a;
    size_t size = sizeof(Args);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &copy, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    // a failed launch is reported like a failed <<< >>> launch: through hipGetLastError, which every C-ABI entry point checks after
    // its launcher returns (hipModuleLaunchKernel hands its status back instead of latching it: a code object loaded on another
    // device than the current one, an unresolved function)
    const hipError_t e = hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, (unsigned)lds, st, nullptr, extra);
    if (e != hipSuccess) g_module_launch_error = e;
}
inline const ModuleUnit* unit_of(const SpecEntry* self) { return static_cast<const ModuleUnit*>(self->module_ctx); }
inline unsigned blocks_of(int64_t n) { return (unsigned)((n + SPEC_BLOCK - 1) / SPEC_BLOCK); }
inline int scene_switch(const ModuleUnit* u, const SpecArgs& a) {
    return u->fast_switch ? (scene_is_fast(a.C) ? 1 : 0) : (scene_is_general(a.C) ? 1 : 0);
}
void mod_rollout(const SpecEntry* self, const SpecArgs& a, int bi, hipStream_t st) {
    const ModuleUnit* u = unit_of(self);
    const int io = a.io_f16 == TRK_IO_F16 ? 1 : (a.io_f16 == TRK_IO_F16_G32 ? 2 : 0);
    mod_launch(u->rollout[io][u->chunked && a.link_pos ? 1 : 0][scene_switch(u, a)][bi ? 1 : 0], blocks_of(a.n), SPEC_BLOCK, 0, a, st);
}
int mod_gp(const SpecEntry* self, const SpecArgs& a, int bi, hipStream_t st) {
    const ModuleUnit* u = unit_of(self);
    const int io = a.io_f16 == TRK_IO_F16 ? 1 : (a.io_f16 == TRK_IO_F16_G32 ? 2 : 0);
    mod_launch(u->gp[io][scene_switch(u, a)][bi ? 1 : 0], blocks_of(a.n), SPEC_BLOCK, 0, a, st);
    return 0;
}
#define TRK_MOD_SIMPLE(NAME, FIELD) \
    void NAME(const SpecEntry* self, const SpecArgs& a, int bi, hipStream_t st) { \
        mod_launch(unit_of(self)->FIELD[bi ? 1 : 0], blocks_of(a.n), SPEC_BLOCK, 0, a, st); }
TRK_MOD_SIMPLE(mod_posbwd, posbwd)
TRK_MOD_SIMPLE(mod_coll, coll)
TRK_MOD_SIMPLE(mod_fkh, fkh)
TRK_MOD_SIMPLE(mod_fkhbwd, fkhbwd)
TRK_MOD_SIMPLE(mod_fk1, fk1)
#undef TRK_MOD_SIMPLE
void mod_fields(const SpecEntry* self, const SpecArgs& a, int, hipStream_t st) {
    const ModuleUnit* u = unit_of(self);
    mod_launch(a.coll_out ? u->collf : u->fields, blocks_of(a.n), SPEC_BLOCK, 0, a, st);
}
void mod_ik(const SpecEntry* self, const IkArgs& a, int bi, hipStream_t st) {
    mod_launch(unit_of(self)->ik[bi ? 1 : 0], blocks_of(a.n), SPEC_BLOCK, 0, a, st);
}
void mod_ikgn(const SpecEntry* self, const IkGnArgs& a, int bi, hipStream_t st) {
    mod_launch(unit_of(self)->ikgn[bi ? 1 : 0], blocks_of(a.n), SPEC_BLOCK, 0, a, st);
}
void mod_jac(const SpecEntry* self, const SpecArgs& a, int bi, hipStream_t st) {
    const ModuleUnit* u = unit_of(self);
    size_t lds;
    if (u->jac_direct) lds = sizeof(float) * (size_t)TRK_WAVE * 6 * u->D;
    else {
        const int rstride = (6 * a.jac_n_cols + 3) | 1;
        lds = sizeof(float) * ((size_t)TRK_WAVE * (rstride > u->D ? rstride : u->D) + TRK_MAX_DOFS);
    }
    mod_launch(u->jac[bi ? 1 : 0], (unsigned)((a.n + TRK_WAVE - 1) / TRK_WAVE), TRK_WAVE, lds, a, st);
}
// an attached-point unit's fused rollout: the choice its generated launcher makes (codegen._points_entry_lines)
void mod_points_rollout(const SpecEntry* self, const SpecArgs& a, int bi, hipStream_t st) {
    const ModuleUnit* u = unit_of(self);
    const bool fast = scene_is_fast(a.C) || a.w.w_obj == 0.0f;
    mod_launch(u->prollout[fast ? 1 : 0][bi ? 1 : 0], blocks_of(a.n), SPEC_BLOCK, 0, a, st);
}
}  // namespace

int trk_spec_register_module(const TrkModuleUnitDesc* d) {
    if (!d || !d->ident || !d->code || d->code_size == 0 || d->n_kernels < 1 || !d->name_exprs || !d->lowered_names || d->n_dofs < 0 ||
        d->n_links < 1 || d->n_obj_links < 0 || d->n_self_pairs < 0 || d->n_virtual < 0)
        return fail(TRK_ERR_INVALID_ARG, "trk_spec_register_module: bad descriptor");
    if (d->spec_abi_version != TRK_SPEC_ABI_VERSION || d->sizeof_args != sizeof(SpecArgs) + sizeof(IkArgs) + sizeof(IkGnArgs) ||
        d->sizeof_cost_hdr != sizeof(DevCostHdr))
        return fail(TRK_ERR_INVALID_ARG, "trk_spec_register_module: the code object was built against another SpecArgs / DevCostHdr layout");
    ModuleUnit* u = new (std::nothrow) ModuleUnit();
    if (!u) return fail(TRK_ERR_HIP, "trk_spec_register_module: out of memory");
    hipError_t e = hipModuleLoadData(&u->mod, d->code);
    if (e != hipSuccess) { delete u; return hip_fail(e, "trk_spec_register_module: hipModuleLoadData"); }
    u->ident = d->ident;
    u->chunked = d->chunked != 0; u->fast_switch = d->fast_switch != 0; u->jac_direct = d->jac_direct != 0; u->D = d->n_dofs;
    u->obj.assign(d->obj_link_idx, d->obj_link_idx + d->n_obj_links);
    u->pairs.assign(d->self_pairs, d->self_pairs + 2 * d->n_self_pairs);
    if (d->n_virtual) { u->vsrc.assign(d->virtual_src, d->virtual_src + 2 * d->n_virtual); u->vw.assign(d->virtual_w, d->virtual_w + 2 * d->n_virtual); }
    const std::string ns = "spec_" + u->ident + "::";
    auto find = [&](const std::string& expr, hipFunction_t* out) -> bool {
        for (int k = 0; k < d->n_kernels; ++k)
            if (expr == d->name_exprs[k]) return hipModuleGetFunction(out, u->mod, d->lowered_names[k]) == hipSuccess;
        return false;
    };
    bool ok = true;
    const char* ios[3] = {"float", "_Float16", "HalfG32"};
    const char* tf[2] = {"false", "true"};
    const char* bs[2] = {"bg", "bi"};
    if (d->n_points > 0) {
        // attached-point unit: fused rollout (FAST / general scene, identity / general base) + the positions' reverse mode
        for (int b = 0; b < 2 && ok; ++b) {
            for (int f = 0; f < 2 && ok; ++f) ok = find(ns + "k_rollout_" + bs[b] + "<" + tf[f] + ", float>", &u->prollout[f][b]);
            ok = ok && find(ns + "k_posbwd_" + bs[b], &u->posbwd[b]);
        }
        if (!ok) {
            (void)hipModuleUnload(u->mod);
            delete u;
            return fail(TRK_ERR_INVALID_ARG, "trk_spec_register_module: a kernel of the attached-point unit is missing from the code object");
        }
        SpecEntry& E = u->entry;
        E.spec_abi_version = TRK_SPEC_ABI_VERSION;
        E.sizeof_args = (uint32_t)(sizeof(SpecArgs) + sizeof(IkArgs) + sizeof(IkGnArgs));
        E.sizeof_entry = (uint32_t)sizeof(SpecEntry); E.sizeof_cost_hdr = (uint32_t)sizeof(DevCostHdr);
        E.model_hash = d->model_hash; E.n_links = d->n_links; E.n_dofs = d->n_dofs;
        E.n_obj_links = d->n_obj_links; E.obj_link_idx = u->obj.data();
        E.n_self_pairs = d->n_self_pairs; E.self_pairs = u->pairs.data();
        E.ee_link = d->ee_link; E.ee2_link = d->ee2_link; E.name = u->ident.c_str();
        E.n_points = d->n_points; E.points_hash = d->points_hash;
        E.launch = mod_points_rollout; E.launch_posbwd = mod_posbwd;
        E.module_ctx = u;
        module_units().push_back(u);
        return trk_spec_register(&u->entry);
    }
    for (int b = 0; b < 2 && ok; ++b) {
        for (int io = 0; io < 3 && ok; ++io)
            for (int sw = 0; sw < 2 && ok; ++sw) {
                for (int pos = 0; pos < (u->chunked ? 2 : 1) && ok; ++pos) {
                    const std::string sws = u->chunked ? std::string(tf[pos]) + ", " + tf[sw] : std::string(tf[sw]);
                    ok = find(ns + "k_rollout_" + bs[b] + "<" + ios[io] + ", " + sws + ">", &u->rollout[io][pos][sw][b]);
                }
                if (ok && d->gp_ok) ok = find(ns + "k_rollout_gpt_" + bs[b] + "<" + ios[io] + ", " + tf[sw] + ">", &u->gp[io][sw][b]);
            }
        ok = ok && find(ns + "k_posbwd_" + bs[b], &u->posbwd[b]) && find(ns + "k_coll_" + bs[b], &u->coll[b]) &&
             find(ns + "k_fkh_" + bs[b], &u->fkh[b]) && find(ns + "k_fk1_" + bs[b], &u->fk1[b]);
        if (ok && d->fkhbwd_ok) ok = find(ns + "k_fkhbwd_" + bs[b], &u->fkhbwd[b]);
        if (ok && d->ik_ok) ok = find(ns + "k_ik_" + bs[b], &u->ik[b]);
        if (ok && d->ikgn_ok) ok = find(ns + "k_ikgn_" + bs[b], &u->ikgn[b]);
        if (ok && d->jac_ok) ok = find(ns + "k_jac_" + bs[b], &u->jac[b]);
    }
    if (ok && d->fields_ok) ok = find(ns + "k_fields", &u->fields) && find(ns + "k_collf", &u->collf);
    if (!ok) {
        (void)hipModuleUnload(u->mod);
        delete u;
        return fail(TRK_ERR_INVALID_ARG, "trk_spec_register_module: a kernel of the unit is missing from the code object");
    }
    SpecEntry& E = u->entry;
    E.spec_abi_version = TRK_SPEC_ABI_VERSION;
    E.sizeof_args = (uint32_t)(sizeof(SpecArgs) + sizeof(IkArgs) + sizeof(IkGnArgs));
    E.sizeof_entry = (uint32_t)sizeof(SpecEntry); E.sizeof_cost_hdr = (uint32_t)sizeof(DevCostHdr);
    E.model_hash = d->model_hash; E.n_links = d->n_links; E.n_dofs = d->n_dofs;
    E.n_obj_links = d->n_obj_links; E.obj_link_idx = u->obj.data();
    E.n_self_pairs = d->n_self_pairs; E.self_pairs = u->pairs.data();
    E.ee_link = d->ee_link; E.ee2_link = d->ee2_link; E.name = u->ident.c_str();
    E.n_points = 0; E.points_hash = 0;
    E.n_virtual = d->n_virtual; E.virtual_src = u->vsrc.data(); E.virtual_w = u->vw.data();
    E.launch = mod_rollout; E.launch_posbwd = mod_posbwd; E.launch_coll = mod_coll; E.launch_fkh = mod_fkh; E.launch_fk1 = mod_fk1;
    E.launch_fkhbwd = d->fkhbwd_ok ? mod_fkhbwd : nullptr;
    E.launch_jac = d->jac_ok ? mod_jac : nullptr;
    E.launch_ik = d->ik_ok ? mod_ik : nullptr;
    E.launch_ikgn = d->ikgn_ok ? mod_ikgn : nullptr;
    E.launch_fields = d->fields_ok ? mod_fields : nullptr;
    E.launch_gp = d->gp_ok ? mod_gp : nullptr;
    E.module_ctx = u;
    module_units().push_back(u);
    return trk_spec_register(&u->entry);
}

struct TrkModel {
    DevModelHdr hdr;
    std::vector<DevLink> links;     // by position
    std::vector<int32_t> fin;
    std::vector<int32_t> joint_list_idx;   // by link
    DevLink* d_links = nullptr;
    int32_t* d_fin = nullptr;
    int32_t* d_dofs = nullptr;             // [D] {pos, subtree end, joint type, pad}
    std::vector<int32_t> pos_of_link;      // link (file index) -> pre-order position
    bool unsupported = false;
    mutable const SpecEntry* spec = nullptr;   // any generated unit for these tables (FK-only entry points use it)
    bool spec_enabled = true;
    uint64_t hash = 0;                   // model_hash of the tables
};

// Generated units may be loaded AFTER the model was created (torch_robotics_amd/jit.py dlopens them), so look again
// while there is none.  Several units can exist for one model (different collision templates).
static const SpecEntry* model_spec(const TrkModel* m) {
    if (!m->spec) m->spec = trk_spec_find(m->hash, m->hdr.n_links, m->hdr.n_dofs);
    return m->spec;
}
static const SpecEntry* model_spec_for(const TrkModel* m, const TrkCostModel* cm, const TrkRolloutWeights* w);

struct TrkPointSet {
    DevPointSet dev;
    void* d_blob = nullptr;
    const TrkModel* model = nullptr;
    mutable const SpecEntry* spec = nullptr;   // any generated unit with exactly this point set baked in (positions-only launches)
    uint64_t hash = 0;                   // FNV-1a over (n_points, point_link, point_offset)
};

struct TrkCostModel {
    DevCostHdr hdr;
    void* d_blob = nullptr;
    float4* d_cells = nullptr;           // the voxel grid as (gx, gy, gz, sdf) records (a copy of the caller's two arrays)
    std::vector<int32_t> obj_link_idx;   // host copies, to match a specialised kernel's baked link sets
    std::vector<int32_t> self_pairs;     // mapped to link indices
    std::vector<int32_t> virtual_src;    // host copies of the interpolated-column table (a generated unit must bake the same one)
    std::vector<float> virtual_w;
    bool spec_enabled = true;            // trk_cost_model_enable_specialized: may trk_cost_fields use a generated unit's field kernel
};

static bool spec_matches(const SpecEntry* e, const TrkCostModel* cm, const TrkRolloutWeights* w) {
    // the single-link self distance exists only in the table-driven kernels; interpolated (virtual) columns need a unit that
    // bakes exactly this table (attached-point units bake none)
    if (cm->hdr.self_single && w->w_self != 0.0f) return false;
    if (cm->hdr.n_virtual != e->n_virtual) return false;
    if (e->n_virtual > 0 && (!std::equal(cm->virtual_src.begin(), cm->virtual_src.end(), e->virtual_src) ||
                             std::memcmp(cm->virtual_w.data(), e->virtual_w, sizeof(float) * 2 * e->n_virtual) != 0))
        return false;
    bool ok = true;
    if (w->w_obj != 0.0f || w->w_ws != 0.0f)
        ok = ok && (int)cm->obj_link_idx.size() == e->n_obj_links &&
             std::equal(cm->obj_link_idx.begin(), cm->obj_link_idx.end(), e->obj_link_idx);
    if (w->w_self != 0.0f)
        ok = ok && (int)cm->self_pairs.size() == 2 * e->n_self_pairs &&
             std::equal(cm->self_pairs.begin(), cm->self_pairs.end(), e->self_pairs);
    if (w->w_ee != 0.0f) ok = ok && cm->hdr.ee_link == e->ee_link && cm->hdr.ee2_link == e->ee2_link;
    return ok;
}
// An object field over an EMPTY scene (no objects, no grid) contributes nothing: the table-driven kernels guard their object
// loop with n_objects > 0, the generated objective code does not (its minimum over no object would stay +inf), so the term
// is switched off here -- the result must not depend on which kernel family serves the call.
// The same rule for every VACUOUS term: a self-collision weight on a cost model without self pairs, a workspace weight without a
// workspace box, an EE weight without a tracked link.  Each contributes exactly nothing in the table-driven kernels (empty loops /
// `has_ws` / `ee_link >= 0` guards) -- but a non-zero weight used to demand that the generated unit's baked set equal the cost
// model's EMPTY one, which sent the call to the table-driven kernel: UR10 + Allegro 26.8 -> 743 us with w_self = 1 on a cost model
// without self pairs (profiles/r05_ablation_c4.txt).
static TrkRolloutWeights effective_weights(const TrkCostModel* cm, TrkRolloutWeights w) {
    if (cm->hdr.n_objects == 0 && !cm->hdr.has_grid) w.w_obj = 0.0f;
    if (cm->hdr.n_self_pairs == 0) w.w_self = 0.0f;
    if (!cm->hdr.has_ws) w.w_ws = 0.0f;
    if (cm->hdr.ee_link < 0 && cm->hdr.ee2_link < 0) w.w_ee = 0.0f;
    return w;
}
// the boolean fields' counterpart: a field that has nothing to test is dropped from the mask (its answer is "no collision")
static int32_t effective_fields(const TrkCostModel* cm, int32_t fields) {
    if (cm->hdr.n_self_pairs == 0) fields &= ~TRK_FIELD_SELF;
    if (!cm->hdr.has_ws) fields &= ~TRK_FIELD_WS;
    if (cm->hdr.n_objects == 0 && !cm->hdr.has_grid) fields &= ~TRK_FIELD_OBJECTS;
    return fields;
}
// strict mode: a model with generated units whose call no unit matches is an error, not a 10 - 30 x slower launch
static int strict_refusal(const char* who, const TrkModel* m) {
    if (!strict_specialized() || !m->spec_enabled || !model_spec(m)) return TRK_OK;
    return fail(TRK_ERR_UNSUPPORTED, std::string(who) + ": strict mode (TRK_STRICT_SPECIALIZED / trk_set_strict_specialized): the model has generated "
                                     "kernels but none bakes this cost model's link sets for the non-zero weights -- the call would take the table-driven kernel");
}
// point-set units: like model_spec / model_spec_for (late registration, several templates per point set)
static const SpecEntry* points_spec(const TrkPointSet* ps) {
    if (!ps->spec) ps->spec = trk_spec_find_points(ps->model->hash, ps->hash, ps->dev.n_points);
    return ps->spec;
}
static const SpecEntry* points_spec_for(const TrkPointSet* ps, const TrkCostModel* cm, const TrkRolloutWeights* w) {
    for (const SpecEntry* e : spec_registry())
        if (e->n_points == ps->dev.n_points && e->n_points > 0 && e->model_hash == ps->model->hash &&
            e->points_hash == ps->hash && spec_matches(e, cm, w))
            return e;
    return nullptr;
}
// field kernel on given positions: no kinematic model is involved, the unit is found by its collision template alone
// (columns = all links of the robot the unit was generated for)
static const SpecEntry* fields_spec_for(const TrkCostModel* cm, const TrkRolloutWeights* w) {
    if (!cm->spec_enabled) return nullptr;
    for (const SpecEntry* e : spec_registry())
        if (e->n_points == 0 && e->launch_fields && e->n_links == cm->hdr.n_links_in && spec_matches(e, cm, w))
            return e;
    return nullptr;
}
static const SpecEntry* model_spec_for(const TrkModel* m, const TrkCostModel* cm, const TrkRolloutWeights* w) {
    for (const SpecEntry* e : spec_registry())
        if (e->n_points == 0 && e->model_hash == m->hash && e->n_links == m->hdr.n_links && e->n_dofs == m->hdr.n_dofs &&
            spec_matches(e, cm, w))
            return e;
    return nullptr;
}

extern "C" {

int trk_abi_version(void) { return TRK_ABI_VERSION; }
const char* trk_last_error(void) { return g_err.c_str(); }

int trk_model_create(const TrkKinModelDesc* d, TrkModel** out) {
    if (!d || !out) return fail(TRK_ERR_INVALID_ARG, "trk_model_create: null argument");
    if (d->abi_version != TRK_ABI_VERSION) return fail(TRK_ERR_INVALID_ARG, "trk_model_create: ABI version mismatch");
    const int L = d->n_links, D = d->n_dofs;
    if (L < 1 || L > TRK_MAX_LINKS) return fail(TRK_ERR_UNSUPPORTED, "trk_model_create: n_links out of range [1, 64]");
    if (D < 0 || D > TRK_MAX_DOFS) return fail(TRK_ERR_UNSUPPORTED, "trk_model_create: n_dofs out of range [0, 32]");
    if (d->n_slots < 0 || d->n_slots > TRK_MAX_POSE_SLOTS) return fail(TRK_ERR_UNSUPPORTED, "trk_model_create: too many pose slots");
    const void* ptrs[] = {d->parent, d->joint_type, d->dof_idx, d->R_fixed, d->trans, d->axis, d->rot_axis, d->rot_sign,
                          d->clamp, d->lower, d->upper, d->sf_rot_axis, d->sf_clamp, d->jac_axis, d->joint_list_idx,
                          d->order, d->subtree_end, d->parent_slot, d->store_slot};
    for (const void* p : ptrs) if (!p) return fail(TRK_ERR_INVALID_ARG, "trk_model_create: null table pointer");
    if (d->order[0] != 0) return fail(TRK_ERR_INVALID_ARG, "trk_model_create: order[0] must be the root link 0");
    std::vector<char> seen(L, 0), dof_seen(D > 0 ? D : 1, 0);
    for (int p = 0; p < L; ++p) {
        const int i = d->order[p];
        if (i < 0 || i >= L || seen[i]) return fail(TRK_ERR_INVALID_ARG, "trk_model_create: order is not a permutation");
        seen[i] = 1;
        if (p > 0) {
            const int par = d->parent[i];
            if (par < 0 || par >= L || !seen[par]) return fail(TRK_ERR_INVALID_ARG, "trk_model_create: parent must precede child in order");
            if (d->parent_slot[p] >= d->n_slots || d->store_slot[p] >= d->n_slots)
                return fail(TRK_ERR_INVALID_ARG, "trk_model_create: slot index out of range");
            if (d->parent_slot[p] < 0 && d->order[p - 1] != par)
                return fail(TRK_ERR_INVALID_ARG, "trk_model_create: parent_slot = -1 but the previous position is not the parent");
        }
        if (d->subtree_end[p] <= p || d->subtree_end[p] > L) return fail(TRK_ERR_INVALID_ARG, "trk_model_create: bad subtree_end");
        const int dof = d->dof_idx[i];
        if (d->joint_type[i] != TRK_JOINT_FIXED) {
            if (dof < 0 || dof >= D || dof_seen[dof]) return fail(TRK_ERR_INVALID_ARG, "trk_model_create: bad dof_idx");
            dof_seen[dof] = 1;
        } else if (dof >= 0) return fail(TRK_ERR_INVALID_ARG, "trk_model_create: fixed joint with a DOF");
    }
    int rc = ensure_init();
    if (rc != TRK_OK) return rc;
    TrkModel* m = new (std::nothrow) TrkModel();
    if (!m) return fail(TRK_ERR_HIP, "out of host memory");
    std::memset(&m->hdr, 0, sizeof(m->hdr));
    m->hdr.n_links = L; m->hdr.n_dofs = D; m->hdr.n_slots = d->n_slots;
    std::memcpy(m->hdr.base_R, d->base_R, sizeof(float) * 9);
    std::memcpy(m->hdr.base_t, d->base_t, sizeof(float) * 3);
    m->links.resize(L);
    m->joint_list_idx.assign(d->joint_list_idx, d->joint_list_idx + L);
    m->pos_of_link.assign(L, 0);
    for (int p = 0; p < L; ++p) m->pos_of_link[d->order[p]] = p;
    for (int p = 0; p < L; ++p) {
        const int i = d->order[p];
        DevLink& k = m->links[p];
        std::memset(&k, 0, sizeof(k));
        std::memcpy(k.Rf, d->R_fixed + 9 * i, sizeof(float) * 9);
        std::memcpy(k.trans, d->trans + 3 * i, sizeof(float) * 3);
        std::memcpy(k.axis, d->axis + 3 * i, sizeof(float) * 3);
        k.lower = d->lower[i]; k.upper = d->upper[i]; k.rot_sign = d->rot_sign[i];
        k.type = d->joint_type[i]; k.dof = d->dof_idx[i]; k.rot_axis = d->rot_axis[i]; k.clamp = d->clamp[i];
        k.parent_slot = p == 0 ? -1 : d->parent_slot[p]; k.store_slot = d->store_slot[p]; k.link = i;
        k.sf_rot_axis = d->sf_rot_axis[i]; k.sf_clamp = d->sf_clamp[i]; k.jac_axis = d->jac_axis[i];
        if (k.type == TRK_JOINT_UNSUPPORTED || k.type < 0 || k.type > TRK_JOINT_UNSUPPORTED) m->unsupported = true;
        k.fin_begin = (int32_t)m->fin.size();
        for (int j = 0; j <= p; ++j)
            if (m->links[j].dof >= 0 && d->subtree_end[j] == p + 1) m->fin.push_back(m->links[j].dof);
        k.fin_end = (int32_t)m->fin.size();
    }
    if (m->fin.empty()) m->fin.push_back(0);
    std::vector<int32_t> dofs(4 * (D > 0 ? D : 1), 0);
    for (int p = 0; p < L; ++p)
        if (m->links[p].dof >= 0) {
            int32_t* r = dofs.data() + 4 * m->links[p].dof;
            r[0] = p; r[1] = d->subtree_end[p]; r[2] = m->links[p].type; r[3] = 0;
        }
    hipError_t e = hipMalloc(&m->d_links, sizeof(DevLink) * L);
    if (e == hipSuccess) e = hipMalloc(&m->d_fin, sizeof(int32_t) * m->fin.size());
    if (e == hipSuccess) e = hipMalloc(&m->d_dofs, sizeof(int32_t) * dofs.size());
    if (e == hipSuccess) e = hipMemcpy(m->d_dofs, dofs.data(), sizeof(int32_t) * dofs.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_links, m->links.data(), sizeof(DevLink) * L, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_fin, m->fin.data(), sizeof(int32_t) * m->fin.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (m->d_links) (void)hipFree(m->d_links);
        if (m->d_fin) (void)hipFree(m->d_fin);
        if (m->d_dofs) (void)hipFree(m->d_dofs);
        delete m;
        return hip_fail(e, "trk_model_create: device allocation/copy");
    }
    m->hash = model_hash(d);
    m->spec = trk_spec_find(m->hash, L, D);
    live_add(m, 1);
    *out = m;
    return TRK_OK;
}

void trk_model_destroy(TrkModel* m) {
    if (!m) return;
    live_del(m);
    if (m->d_links) (void)hipFree(m->d_links);
    if (m->d_fin) (void)hipFree(m->d_fin);
    if (m->d_dofs) (void)hipFree(m->d_dofs);
    delete m;
}

int trk_model_set_base_pose(TrkModel* m, const float* R9, const float* t3) {
    if (!m || !R9 || !t3) return fail(TRK_ERR_INVALID_ARG, "trk_model_set_base_pose: null argument");
    std::memcpy(m->hdr.base_R, R9, sizeof(float) * 9);
    std::memcpy(m->hdr.base_t, t3, sizeof(float) * 3);
    return TRK_OK;
}

int trk_model_n_links(const TrkModel* m) { return m ? m->hdr.n_links : TRK_ERR_INVALID_ARG; }
int trk_model_n_dofs(const TrkModel* m) { return m ? m->hdr.n_dofs : TRK_ERR_INVALID_ARG; }
int trk_model_is_specialized(const TrkModel* m) { return (m && model_spec(m) && m->spec_enabled) ? 1 : 0; }
int trk_model_enable_specialized(TrkModel* m, int enable) {
    if (!m) return fail(TRK_ERR_INVALID_ARG, "trk_model_enable_specialized: null model");
    m->spec_enabled = enable != 0;
    return TRK_OK;
}

static int make_sel(const TrkModel* m, const int32_t* link_sel, int32_t n_sel, SelMap& sel, int& n_out, const char* who) {
    const int L = m->hdr.n_links;
    for (int k = 0; k < TRK_MAX_LINKS; ++k) sel.col[k] = -1;
    if (!link_sel) {
        for (int k = 0; k < L; ++k) sel.col[k] = k;
        n_out = L;
        return TRK_OK;
    }
    if (n_sel < 1 || n_sel > L) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": n_sel out of range");
    for (int c = 0; c < n_sel; ++c) {
        const int i = link_sel[c];
        if (i < 0 || i >= L) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": link index out of range");
        if (sel.col[i] >= 0) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": duplicate link in link_sel");
        sel.col[i] = c;
    }
    n_out = n_sel;
    return TRK_OK;
}

// SpecArgs for launches that evaluate no objective (all weights zero): every table pointer of the cost header points at a
// small zero-filled device buffer instead of NULL, so a scalar load the compiler moved out of a weight-guarded branch
// reads zeros rather than faulting.
static void* g_zero_blob = nullptr;
static int blank_spec_args(SpecArgs& a) {
    if (!g_zero_blob) {
        hipError_t e = hipMalloc(&g_zero_blob, 4096);
        if (e == hipSuccess) e = hipMemset(g_zero_blob, 0, 4096);
        if (e != hipSuccess) { g_zero_blob = nullptr; return hip_fail(e, "zero table allocation"); }
    }
    std::memset(&a, 0, sizeof(a));
    DevCostHdr& h = a.C;
    h.ee_link = -1; h.ee2_link = -1;
    h.obj_link_idx = static_cast<const int32_t*>(g_zero_blob); h.obj_link_margin = static_cast<const float*>(g_zero_blob);
    h.objects = static_cast<const DevObj*>(g_zero_blob); h.prims = static_cast<const DevPrim*>(g_zero_blob);
    h.self_pairs = static_cast<const int32_t*>(g_zero_blob); h.self_margin = static_cast<const float*>(g_zero_blob);
    h.spheres = static_cast<const float4*>(g_zero_blob); h.spheres_sel = static_cast<const float4*>(g_zero_blob);
    h.box_objects = static_cast<const int32_t*>(g_zero_blob);
    h.sphere_pairs = static_cast<const float*>(g_zero_blob);
    h.virtual_src = static_cast<const int32_t*>(g_zero_blob); h.virtual_w = static_cast<const float*>(g_zero_blob);
    return TRK_OK;
}

static int base_is_identity(const TrkModel* m) {
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z[3] = {0, 0, 0};
    return std::memcmp(m->hdr.base_R, I, sizeof(I)) == 0 && std::memcmp(m->hdr.base_t, Z, sizeof(Z)) == 0;
}
// the generated kernels produce / consume all links in file order
static bool spec_all_links(const TrkModel* m, const SelMap& sel, int ns) {
    if (!model_spec(m) || !m->spec_enabled || ns != m->hdr.n_links) return false;
    for (int k = 0; k < ns; ++k) if (sel.col[k] != k) return false;
    return true;
}

static int check_model(const TrkModel* m, const char* who) {
    if (!m) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": null model");
    if (m->unsupported) return fail(TRK_ERR_UNSUPPORTED, std::string(who) + ": model has a joint type other than fixed/revolute/continuous/prismatic");
    return TRK_OK;
}

static int fk_fwd(int mode, const TrkModel* m, const float* q, int64_t n, const int32_t* link_sel, int32_t n_sel,
                  float* out, trk_stream_t stream, const char* who) {
    int rc = check_model(m, who);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!out || (!q && m->hdr.n_dofs > 0)))) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": bad q/out/n");
    SelMap sel; int ns;
    rc = make_sel(m, link_sel, n_sel, sel, ns, who);
    if (rc) return rc;
    if (n == 0) return TRK_OK;
    if (mode == 1 && spec_all_links(m, sel, ns)) {
        // generated kernel, positions-only exit (gq == nullptr): same FK code as the fused rollout
        SpecArgs a{};
        rc = blank_spec_args(a);
        if (rc) return rc;
        std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
        std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
        a.q = q; a.n = n; a.link_pos = out;
        m->spec->launch(m->spec, a, base_is_identity(m), (hipStream_t)stream);
        TRK_HIP(last_launch_error());
        return TRK_OK;
    }
    if (mode == 0 && ns == 1 && model_spec(m) && m->spec_enabled && m->spec->launch_fk1) {
        // generated kernel: the unrolled stateless walk up to the one selected link
        SpecArgs a{};
        rc = blank_spec_args(a);
        if (rc) return rc;
        std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
        std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
        a.q = q; a.n = n; a.fk_H = out;
        a.jac_link = -1; a.jac_p_end = 1;
        for (int l = 0; l < m->hdr.n_links; ++l) if (sel.col[l] == 0) a.jac_link = l;
        for (int p = 0; p < m->hdr.n_links; ++p) if (m->links[p].link == a.jac_link) a.jac_p_end = p + 1;
        if (a.jac_link >= 0) {
            m->spec->launch_fk1(m->spec, a, base_is_identity(m), (hipStream_t)stream);
            TRK_HIP(last_launch_error());
            return TRK_OK;
        }
    }
    if (mode == 0 && spec_all_links(m, sel, ns) && m->spec->launch_fkh) {
        // generated kernel: the unrolled stateless walk, every link's 4x4 streamed out as it exists
        SpecArgs a{};
        rc = blank_spec_args(a);
        if (rc) return rc;
        std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
        std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
        a.q = q; a.n = n; a.fk_H = out;
        m->spec->launch_fkh(m->spec, a, base_is_identity(m), (hipStream_t)stream);
        TRK_HIP(last_launch_error());
        return TRK_OK;
    }
    trk_launch_fk_forward(mode, m->hdr, m->d_links, sel, ns, q, n, out, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_fk_forward(const TrkModel* m, const float* q, int64_t n, const int32_t* link_sel, int32_t n_sel, float* H_out, trk_stream_t stream) {
    return fk_fwd(0, m, q, n, link_sel, n_sel, H_out, stream, "trk_fk_forward");
}
int trk_fk_positions(const TrkModel* m, const float* q, int64_t n, const int32_t* link_sel, int32_t n_sel, float* pos_out, trk_stream_t stream) {
    return fk_fwd(1, m, q, n, link_sel, n_sel, pos_out, stream, "trk_fk_positions");
}

static int fk_bwd(int mode, const TrkModel* m, const float* q, const float* gin, int64_t n, const int32_t* link_sel,
                  int32_t n_sel, float* gq, trk_stream_t stream, const char* who) {
    int rc = check_model(m, who);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!gin || ((!q || !gq) && m->hdr.n_dofs > 0)))) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": bad q/g/n");
    SelMap sel; int ns;
    rc = make_sel(m, link_sel, n_sel, sel, ns, who);
    if (rc) return rc;
    if (n == 0 || m->hdr.n_dofs == 0) return TRK_OK;
    if (mode == 0 && spec_all_links(m, sel, ns) && m->spec->launch_fkhbwd) {
        SpecArgs a{};
        rc = blank_spec_args(a);
        if (rc) return rc;
        std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
        std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
        a.q = q; a.n = n; a.fk_H = const_cast<float*>(gin); a.gq = gq;
        m->spec->launch_fkhbwd(m->spec, a, base_is_identity(m), (hipStream_t)stream);
        TRK_HIP(last_launch_error());
        return TRK_OK;
    }
    if (mode == 1 && spec_all_links(m, sel, ns) && m->spec->launch_posbwd) {
        SpecArgs a{};
        rc = blank_spec_args(a);
        if (rc) return rc;
        std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
        std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
        a.q = q; a.n = n; a.link_pos = const_cast<float*>(gin); a.gq = gq;
        m->spec->launch_posbwd(m->spec, a, base_is_identity(m), (hipStream_t)stream);
        TRK_HIP(last_launch_error());
        return TRK_OK;
    }
    SelMap selp;
    for (int k = 0; k < TRK_MAX_LINKS; ++k) selp.col[k] = -1;
    for (int p = 0; p < m->hdr.n_links; ++p) selp.col[p] = sel.col[m->links[p].link];
    trk_launch_fk_backward(mode, m->hdr, m->d_links, m->d_fin, sel, selp, ns, q, gin, n, gq, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_fk_backward(const TrkModel* m, const float* q, const float* gH, int64_t n, const int32_t* link_sel, int32_t n_sel, float* gq, trk_stream_t stream) {
    return fk_bwd(0, m, q, gH, n, link_sel, n_sel, gq, stream, "trk_fk_backward");
}
int trk_fk_positions_backward(const TrkModel* m, const float* q, const float* gpos, int64_t n, const int32_t* link_sel, int32_t n_sel, float* gq, trk_stream_t stream) {
    return fk_bwd(1, m, q, gpos, n, link_sel, n_sel, gq, stream, "trk_fk_positions_backward");
}

int trk_point_set_create(const TrkModel* m, const int32_t* point_link, const float* point_offset, int32_t n_points,
                         TrkPointSet** out) {
    if (!m || !out || !point_link || !point_offset) return fail(TRK_ERR_INVALID_ARG, "trk_point_set_create: null argument");
    if (n_points < 1 || n_points > TRK_MAX_POINTS) return fail(TRK_ERR_UNSUPPORTED, "trk_point_set_create: n_points out of range [1, 192]");
    const int L = m->hdr.n_links;
    for (int k = 0; k < n_points; ++k)
        if (point_link[k] < 0 || point_link[k] >= L) return fail(TRK_ERR_INVALID_ARG, "trk_point_set_create: link index out of range");
    // sort by the pre-order position of the owning link (stable: keeps the caller's order inside a link)
    std::vector<int32_t> begin(L + 1, 0);
    for (int k = 0; k < n_points; ++k) ++begin[m->pos_of_link[point_link[k]] + 1];
    for (int p = 0; p < L; ++p) begin[p + 1] += begin[p];
    std::vector<DevPoint> pts(n_points);
    std::vector<int32_t> fill(begin.begin(), begin.end() - 1);
    for (int k = 0; k < n_points; ++k) {
        DevPoint& d = pts[fill[m->pos_of_link[point_link[k]]]++];
        std::memcpy(d.off, point_offset + 3 * k, sizeof(float) * 3);
        d.col = k;
    }
    const size_t o_begin = sizeof(DevPoint) * n_points;
    const size_t total = o_begin + sizeof(int32_t) * (L + 1);
    TrkPointSet* ps = new (std::nothrow) TrkPointSet();
    if (!ps) return fail(TRK_ERR_HIP, "out of host memory");
    hipError_t e = hipMalloc(&ps->d_blob, total);
    if (e == hipSuccess) e = hipMemcpy(ps->d_blob, pts.data(), o_begin, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(static_cast<char*>(ps->d_blob) + o_begin, begin.data(), sizeof(int32_t) * (L + 1), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (ps->d_blob) (void)hipFree(ps->d_blob);
        delete ps;
        return hip_fail(e, "trk_point_set_create: device allocation/copy");
    }
    ps->dev.pts = static_cast<const DevPoint*>(ps->d_blob);
    ps->dev.begin = reinterpret_cast<const int32_t*>(static_cast<char*>(ps->d_blob) + o_begin);
    ps->dev.n_points = n_points; ps->dev._pad = 0;
    ps->model = m;
    {   // must hash the same bytes as torch_robotics_amd/codegen.py: points_hash
        uint64_t h = 0xcbf29ce484222325ull;
        h = fnv1a(h, &n_points, sizeof(int32_t));
        h = fnv1a(h, point_link, sizeof(int32_t) * n_points);
        h = fnv1a(h, point_offset, sizeof(float) * 3 * n_points);
        ps->hash = h;
        ps->spec = trk_spec_find_points(m->hash, h, n_points);
    }
    live_add(ps, 3);
    *out = ps;
    return TRK_OK;
}

void trk_point_set_destroy(TrkPointSet* ps) {
    if (!ps) return;
    live_del(ps);
    if (ps->d_blob) (void)hipFree(ps->d_blob);
    delete ps;
}

int trk_point_set_size(const TrkPointSet* ps) { return ps ? ps->dev.n_points : TRK_ERR_INVALID_ARG; }
int trk_point_set_is_specialized(const TrkPointSet* ps) { return (ps && points_spec(ps) && ps->model->spec_enabled) ? 1 : 0; }

static const size_t kMaxLds = 160 * 1024;

int trk_fk_points(const TrkModel* m, const TrkPointSet* ps, const float* q, int64_t n, float* pos_out, trk_stream_t stream) {
    int rc = check_model(m, "trk_fk_points");
    if (rc) return rc;
    if (!ps || ps->model != m) return fail(TRK_ERR_INVALID_ARG, "trk_fk_points: point set does not belong to this model");
    if (n < 0 || (n > 0 && (!pos_out || (!q && m->hdr.n_dofs > 0)))) return fail(TRK_ERR_INVALID_ARG, "trk_fk_points: bad q/out/n");
    if (trk_lds_fk_points(m->hdr, ps->dev.n_points, false) > kMaxLds) return fail(TRK_ERR_UNSUPPORTED, "trk_fk_points: point tile exceeds the 160 KiB LDS");
    if (n == 0) return TRK_OK;
    if (points_spec(ps) && m->spec_enabled && (reinterpret_cast<uintptr_t>(pos_out) & 15) == 0) {
        // generated kernel with this point set baked in, all weights zero and no gradient output: FK + positions only
        SpecArgs a{};
        rc = blank_spec_args(a);
        if (rc) return rc;
        std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
        std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
        a.q = q; a.n = n; a.link_pos = pos_out;
        ps->spec->launch(ps->spec, a, base_is_identity(m), (hipStream_t)stream);
        TRK_HIP(last_launch_error());
        return TRK_OK;
    }
    trk_launch_fk_points(m->hdr, m->d_links, ps->dev, q, n, pos_out, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_fk_points_backward(const TrkModel* m, const TrkPointSet* ps, const float* q, const float* gpos, int64_t n, float* gq,
                           trk_stream_t stream) {
    int rc = check_model(m, "trk_fk_points_backward");
    if (rc) return rc;
    if (!ps || ps->model != m) return fail(TRK_ERR_INVALID_ARG, "trk_fk_points_backward: point set does not belong to this model");
    if (n < 0 || (n > 0 && (!gpos || ((!q || !gq) && m->hdr.n_dofs > 0)))) return fail(TRK_ERR_INVALID_ARG, "trk_fk_points_backward: bad q/g/n");
    if (trk_lds_fk_points(m->hdr, ps->dev.n_points, true) > kMaxLds) return fail(TRK_ERR_UNSUPPORTED, "trk_fk_points_backward: point tile exceeds the 160 KiB LDS");
    if (n == 0 || m->hdr.n_dofs == 0) return TRK_OK;
    if (points_spec(ps) && ps->spec->launch_posbwd && m->spec_enabled && (reinterpret_cast<uintptr_t>(gpos) & 15) == 0) {
        SpecArgs a{};
        rc = blank_spec_args(a);
        if (rc) return rc;
        std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
        std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
        a.q = q; a.n = n; a.link_pos = const_cast<float*>(gpos); a.gq = gq;
        ps->spec->launch_posbwd(ps->spec, a, base_is_identity(m), (hipStream_t)stream);
        TRK_HIP(last_launch_error());
        return TRK_OK;
    }
    trk_launch_fk_points_backward(m->hdr, m->d_links, m->d_fin, ps->dev, q, gpos, n, gq, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_fk_jacobian(const TrkModel* m, const float* q, const float* qd, int64_t n, int32_t link, float* pos, float* quat,
                    float* lin_jac, float* ang_jac, float* vel_lin, float* vel_ang, trk_stream_t stream) {
    int rc = check_model(m, "trk_fk_jacobian");
    if (rc) return rc;
    if (link < 0 || link >= m->hdr.n_links) return fail(TRK_ERR_INVALID_ARG, "trk_fk_jacobian: link out of range");
    if (n < 0 || (n > 0 && (!q || !pos || !quat || !lin_jac || !ang_jac))) return fail(TRK_ERR_INVALID_ARG, "trk_fk_jacobian: null argument");
    if (n == 0) return TRK_OK;
    if (m->spec_enabled && m->spec && m->spec->launch_jac && !vel_lin && !vel_ang) {
        // generated kernel: the stateful walk unrolled with the URDF constants folded; the link velocities stay table-driven
        SpecArgs a{};
        rc = blank_spec_args(a);
        if (rc) return rc;
        std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
        std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
        a.q = q; a.n = n; a.jac_link = link; a.jac_joint_idx = m->joint_list_idx[link];
        // which DOFs get a column and where the walk may stop (same rule as trk_launch_fk_jacobian: robot_tree.py:239-244)
        a.jac_p_end = 1; a.jac_n_cols = 0;
        for (int d = 0; d < TRK_MAX_DOFS; ++d) a.jac_slot[d] = -1;
        for (int p = 0; p < m->hdr.n_links; ++p) {
            const DevLink& Lk = m->links[p];
            if (Lk.link == link) a.jac_p_end = std::max(a.jac_p_end, p + 1);
            if (Lk.dof >= 0 && (Lk.link - 1) <= a.jac_joint_idx && Lk.jac_axis >= 0) {
                a.jac_slot[Lk.dof] = (int8_t)a.jac_n_cols++;
                a.jac_p_end = std::max(a.jac_p_end, p + 1);
            }
        }
        a.jac_pos = pos; a.jac_quat = quat; a.jac_lin = lin_jac; a.jac_ang = ang_jac;
        m->spec->launch_jac(m->spec, a, base_is_identity(m), (hipStream_t)stream);
        TRK_HIP(last_launch_error());
        return TRK_OK;
    }
    trk_launch_fk_jacobian(m->hdr, m->d_links, m->links.data(), q, qd, n, link, m->joint_list_idx[link], pos, quat, lin_jac, ang_jac,
                           vel_lin, vel_ang, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_fk_analytic_jacobian(const TrkModel* m, const float* q, int64_t n, float* J, trk_stream_t stream) {
    int rc = check_model(m, "trk_fk_analytic_jacobian");
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!J || (!q && m->hdr.n_dofs > 0)))) return fail(TRK_ERR_INVALID_ARG, "trk_fk_analytic_jacobian: bad q/J/n");
    if (n == 0 || m->hdr.n_dofs == 0) return TRK_OK;
    if (m->spec_enabled && model_spec(m) && m->spec->launch_ajac) {
        SpecArgs a{};
        std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
        std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
        a.q = q; a.n = n; a.jac_lin = J;
        m->spec->launch_ajac(m->spec, a, base_is_identity(m), (hipStream_t)stream);
        TRK_HIP(last_launch_error());
        return TRK_OK;
    }
    trk_launch_fk_analytic_jacobian(m->hdr, m->d_links, m->d_dofs, q, n, J, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_ik_steps(const TrkModel* m, int32_t link, const float* H_target, int32_t per_sample_target, const float* lower,
                 const float* upper, float w_joint_limits, float se3_eps, float lr, int32_t first_step, int32_t n_steps, int64_t n,
                 float* q, float* adam_m, float* adam_v, float* loss, uint8_t* valid, trk_stream_t stream) {
    int rc = check_model(m, "trk_ik_steps");
    if (rc) return rc;
    if (link < 0 || link >= m->hdr.n_links) return fail(TRK_ERR_INVALID_ARG, "trk_ik_steps: link out of range");
    if (n < 0 || first_step < 1 || n_steps < 1 || !H_target || !lower || !upper || (n > 0 && !q) || (lr > 0.0f && (!adam_m || !adam_v)))
        return fail(TRK_ERR_INVALID_ARG, "trk_ik_steps: bad argument");
    if (lr <= 0.0f && n_steps != 1) return fail(TRK_ERR_INVALID_ARG, "trk_ik_steps: lr = 0 only evaluates, n_steps must be 1");
    if (n == 0 || m->hdr.n_dofs == 0) return TRK_OK;
    // generated kernel (configurations and Adam state in registers) when the target is the link the unit tracks
    const SpecEntry* gen = (model_spec(m) && m->spec_enabled && m->spec->launch_ik && m->spec->ee_link == link) ? m->spec : nullptr;
    // at most TRK_IK_MAX_STEPS iterations per launch; loss / valid come from the first launch (q as the caller passed it)
    for (int32_t done = 0; done < n_steps; done += TRK_IK_MAX_STEPS) {
        const int32_t k = std::min<int32_t>(TRK_IK_MAX_STEPS, n_steps - done);
        IkSchedule sched{};
        for (int32_t i = 0; i < k; ++i) {
            const float t = (float)(first_step + done + i);
            sched.bc1[i] = 1.0f - std::pow(0.9f, t);
            sched.rsqrt_bc2[i] = 1.0f / std::sqrt(1.0f - std::pow(0.999f, t));
        }
        if (gen) {
            IkArgs a{};
            std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
            std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
            a.H_target = H_target; a.per_sample = per_sample_target; a.n_steps = k; a.lower = lower; a.upper = upper;
            a.w_jl = w_joint_limits; a.se3_eps = se3_eps; a.lr = lr; a.sched = sched; a.n = n;
            a.q = q; a.adam_m = adam_m; a.adam_v = adam_v;
            a.loss = done == 0 ? loss : nullptr; a.valid = done == 0 ? valid : nullptr;
            gen->launch_ik(gen, a, base_is_identity(m), (hipStream_t)stream);
            TRK_HIP(last_launch_error());
            continue;
        }
        trk_launch_ik_step(m->hdr, m->d_links, m->d_fin, link, H_target, per_sample_target, lower, upper, w_joint_limits,
                           se3_eps, lr, sched, k, n, q, adam_m, adam_v, done == 0 ? loss : nullptr, done == 0 ? valid : nullptr,
                           (hipStream_t)stream);
        TRK_HIP(last_launch_error());
    }
    return TRK_OK;
}

int trk_ik_gn_steps(const TrkModel* m, int32_t link, const float* H_target, int32_t per_sample_target, const float* lower,
                    const float* upper, float damping, float lm_gain, float step_scale, float se3_eps, int32_t n_steps, int64_t n,
                    float* q, float* err, uint8_t* valid, trk_stream_t stream) {
    int rc = check_model(m, "trk_ik_gn_steps");
    if (rc) return rc;
    if (link < 0 || link >= m->hdr.n_links) return fail(TRK_ERR_INVALID_ARG, "trk_ik_gn_steps: link out of range");
    if (n < 0 || n_steps < 1 || !H_target || !lower || !upper || (n > 0 && !q) || !(damping >= 0.0f) || !(lm_gain >= 0.0f) ||
        !(damping + lm_gain > 0.0f) || !std::isfinite(step_scale))
        return fail(TRK_ERR_INVALID_ARG, "trk_ik_gn_steps: bad argument (damping, lm_gain >= 0 and not both zero)");
    if (n == 0 || m->hdr.n_dofs == 0) return TRK_OK;
    // the Jacobian, the normal equations and their factor live in one lane's registers: a generated kernel only, for the link a
    // unit tracks (any link of any robot up to 9 DOF is one jit.specialize away)
    const SpecEntry* gen = nullptr;
    if (m->spec_enabled)
        for (const SpecEntry* e : spec_registry())
            if (e->n_points == 0 && e->model_hash == m->hash && e->n_links == m->hdr.n_links && e->n_dofs == m->hdr.n_dofs &&
                e->launch_ikgn && e->ee_link == link) { gen = e; break; }
    if (!gen)
        return fail(TRK_ERR_UNSUPPORTED, "trk_ik_gn_steps: no generated unit of this model tracks this link (robots up to 9 DOF: "
                                         "torch_robotics_amd.jit.specialize(kin, obj_links, ee_link=link)); the two-launch form is "
                                         "trk_fk_jacobian + trk_jtj");
    IkGnArgs a{};
    std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
    std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
    a.H_target = H_target; a.per_sample = per_sample_target; a.n_steps = n_steps; a.lower = lower; a.upper = upper;
    a.damping = damping; a.lm_gain = lm_gain; a.step_scale = step_scale; a.se3_eps = se3_eps; a.n = n;
    a.q = q; a.err = err; a.valid = valid;
    gen->launch_ikgn(gen, a, base_is_identity(m), (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_ik_step(const TrkModel* m, int32_t link, const float* H_target, int32_t per_sample_target, const float* lower,
                const float* upper, float w_joint_limits, float se3_eps, float lr, int32_t step, int64_t n, float* q,
                float* adam_m, float* adam_v, float* loss, uint8_t* valid, trk_stream_t stream) {
    return trk_ik_steps(m, link, H_target, per_sample_target, lower, upper, w_joint_limits, se3_eps, lr, step, 1, n, q, adam_m,
                        adam_v, loss, valid, stream);
}

int trk_rotmat_to_quat(const float* R, int64_t n, int32_t stride, int32_t row_pitch, float* quat, trk_stream_t stream) {
    if (n < 0 || stride < 9 || row_pitch < 3 || (n > 0 && (!R || !quat))) return fail(TRK_ERR_INVALID_ARG, "trk_rotmat_to_quat: bad argument");
    if (n == 0) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_rotmat_to_quat(R, n, stride, row_pitch, quat, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_rotation_from(int32_t kind, const float* in, int64_t n, float* R_out, trk_stream_t stream) {
    if (kind < TRK_ROT_X || kind > TRK_ROT_QUAT_WXYZ || n < 0 || (n > 0 && (!in || !R_out))) return fail(TRK_ERR_INVALID_ARG, "trk_rotation_from: bad argument");
    if (n == 0) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_rotation_from(kind, in, n, R_out, nullptr, nullptr, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_rotation_from_backward(int32_t kind, const float* angle, const float* gR, int64_t n, float* gangle, trk_stream_t stream) {
    if (kind < TRK_ROT_X || kind > TRK_ROT_QUAT_WXYZ || n < 0 || (n > 0 && (!angle || !gR || !gangle))) return fail(TRK_ERR_INVALID_ARG, "trk_rotation_from_backward: bad argument");
    if (n == 0) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_rotation_from(kind, angle, n, nullptr, gR, gangle, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

// ------------------------------------------------------------------------------------------------
// Frame algebra (geometrics/frame.py:55-121)
int trk_frame_compose(int32_t op, const float* Ra, const float* ta, int64_t na, const float* Rb, const float* tb, int64_t nb,
                      float* R_out, float* t_out, trk_stream_t stream) {
    if (op < TRK_FRAME_COMPOSE || op > TRK_FRAME_INV_COMPOSE) return fail(TRK_ERR_INVALID_ARG, "trk_frame_compose: unknown op");
    const bool two = op != TRK_FRAME_INVERSE;
    if (na < 0 || (two && nb < 0)) return fail(TRK_ERR_INVALID_ARG, "trk_frame_compose: negative size");
    const int64_t n = two ? std::max(na, nb) : na;
    if (two && na != nb && na != 1 && nb != 1 && n > 0)
        return fail(TRK_ERR_INVALID_ARG, "trk_frame_compose: batch sizes differ and neither is 1");
    if (two && (na == 0 || nb == 0)) return TRK_OK;
    if (n == 0) return TRK_OK;
    if (!Ra || !ta || (two && (!Rb || !tb)) || !R_out || !t_out) return fail(TRK_ERR_INVALID_ARG, "trk_frame_compose: null pointer");
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_frame_compose(op, Ra, ta, two && na == 1 && n > 1, Rb, tb, two && nb == 1 && n > 1, n, R_out, t_out,
                             (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_frame_compose_backward(int32_t op, const float* Ra, const float* ta, const float* Rb, const float* tb, const float* gR,
                               const float* gt, int64_t n, float* gRa, float* gta, float* gRb, float* gtb, trk_stream_t stream) {
    if (op < TRK_FRAME_COMPOSE || op > TRK_FRAME_INV_COMPOSE) return fail(TRK_ERR_INVALID_ARG, "trk_frame_compose_backward: unknown op");
    if (n < 0) return fail(TRK_ERR_INVALID_ARG, "trk_frame_compose_backward: negative size");
    if (n == 0) return TRK_OK;
    const bool two = op != TRK_FRAME_INVERSE;
    if (!Ra || !ta || !gR || !gt || !gRa || !gta || (two && (!Rb || !tb || !gRb || !gtb)))
        return fail(TRK_ERR_INVALID_ARG, "trk_frame_compose_backward: null pointer");
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_frame_compose_bwd(op, Ra, ta, Rb, tb, gR, gt, n, gRa, gta, gRb, gtb, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_frame_transform_points(const float* R, const float* t, int64_t n, const float* points, int32_t P, float* out,
                               trk_stream_t stream) {
    if (n < 0 || P < 0) return fail(TRK_ERR_INVALID_ARG, "trk_frame_transform_points: negative size");
    if (n == 0 || P == 0) return TRK_OK;
    if (!R || !t || !points || !out) return fail(TRK_ERR_INVALID_ARG, "trk_frame_transform_points: null pointer");
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_frame_transform_points(R, t, n, points, P, out, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_frame_transform_points_backward(const float* gout, int64_t n, const float* points, int32_t P, float* gR, float* gt,
                                        trk_stream_t stream) {
    if (n < 0 || P < 0) return fail(TRK_ERR_INVALID_ARG, "trk_frame_transform_points_backward: negative size");
    if (n == 0) return TRK_OK;
    if ((P > 0 && (!gout || !points)) || !gR || !gt) return fail(TRK_ERR_INVALID_ARG, "trk_frame_transform_points_backward: null pointer");
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_frame_transform_points_bwd(gout, n, points, P, gR, gt, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_frame_quat_euler(const float* R, int64_t n, int32_t stride, int32_t row_pitch, float* quat_xyzw, float* euler,
                         trk_stream_t stream) {
    if (n < 0 || stride < 9 || row_pitch < 3 || (n > 0 && !R)) return fail(TRK_ERR_INVALID_ARG, "trk_frame_quat_euler: bad argument");
    if (n == 0 || (!quat_xyzw && !euler)) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_frame_quat_euler(R, n, stride, row_pitch, quat_xyzw, euler, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_frame_quat_euler_backward(const float* R, int64_t n, int32_t stride, int32_t row_pitch, const float* gquat_xyzw,
                                  const float* geuler, float* gR, trk_stream_t stream) {
    if (n < 0 || stride < 9 || row_pitch < 3 || (n > 0 && (!R || !gR))) return fail(TRK_ERR_INVALID_ARG, "trk_frame_quat_euler_backward: bad argument");
    if (n == 0) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_frame_quat_euler_bwd(R, n, stride, row_pitch, gquat_xyzw, geuler, gR, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

// ------------------------------------------------------------------------------------------------
int trk_cost_model_create(const TrkCostModelDesc* d, TrkCostModel** out) {
    if (!d || !out) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: null argument");
    if (d->abi_version != TRK_ABI_VERSION) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: ABI version mismatch");
    const int Lin = d->n_links_in;
    if (Lin < 1 || Lin > 4 * TRK_MAX_LINKS) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: n_links_in out of range");
    if (d->n_obj_links < 0 || d->n_obj_links > TRK_MAX_COLL_LINKS || d->n_objects < 0 || d->n_objects > TRK_MAX_OBJECTS ||
        d->n_prims < 0 || d->n_prims > TRK_MAX_PRIMS || d->n_self_links < 0 || d->n_self_links > TRK_MAX_COLL_LINKS ||
        d->n_self_pairs < 0 || d->n_self_pairs > TRK_MAX_SELF_PAIRS)
        return fail(TRK_ERR_UNSUPPORTED, "trk_cost_model_create: table size out of range");
    if ((d->n_obj_links && (!d->obj_link_idx || !d->obj_link_margin)) || (d->n_objects && !d->objects) ||
        (d->n_prims && !d->prims) || (d->n_self_links && !d->self_link_idx) ||
        (d->n_self_pairs && (!d->self_pairs || !d->self_margin)))
        return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: null table pointer");
    if (d->n_virtual < 0 || d->n_virtual > TRK_MAX_VIRTUAL) return fail(TRK_ERR_UNSUPPORTED, "trk_cost_model_create: n_virtual out of range");
    if (d->n_virtual && (!d->virtual_src || !d->virtual_w)) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: null virtual column table");
    for (int v = 0; v < 2 * d->n_virtual; ++v)
        if (d->virtual_src[v] < 0 || d->virtual_src[v] >= Lin) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: virtual_src must name real columns");
    const int Lcols = Lin + d->n_virtual;        // what the index tables may address
    for (int l = 0; l < d->n_obj_links; ++l)
        if (d->obj_link_idx[l] < 0 || d->obj_link_idx[l] >= Lcols) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: obj_link_idx out of range");
    for (int l = 0; l < d->n_self_links; ++l)
        if (d->self_link_idx[l] < 0 || d->self_link_idx[l] >= Lcols) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: self_link_idx out of range");
    for (int p = 0; p < 2 * d->n_self_pairs; ++p)
        if (d->self_pairs[p] < 0 || d->self_pairs[p] >= d->n_self_links) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: self_pairs out of range");
    if (d->ee_link >= Lin || d->ee2_link >= Lin) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: ee_link out of range");
    if (d->ee2_link >= 0 && d->ee_link < 0) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: ee2_link needs ee_link");
    int n_grid = 0;
    for (int o = 0; o < d->n_objects; ++o) {
        const TrkObject& ob = d->objects[o];
        if (ob.is_grid) { ++n_grid; continue; }
        if (ob.prim_begin < 0 || ob.prim_end < ob.prim_begin || ob.prim_end > d->n_prims)
            return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: object primitive range out of bounds");
        for (int p = ob.prim_begin; p < ob.prim_end; ++p)
            if (d->prims[p].type < TRK_PRIM_SPHERE || d->prims[p].type > TRK_PRIM_SHARP_BOX)
                return fail(TRK_ERR_UNSUPPORTED, "trk_cost_model_create: unknown primitive type");
    }
    if (n_grid > 1 || (n_grid == 1 && (!d->has_grid || !d->grid.sdf || !d->grid.grad)))
        return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_create: grid object without grid data (or more than one grid)");
    int rc = ensure_init();
    if (rc) return rc;

    // one device blob: [obj_link_idx | obj_link_margin | objects | prims | self_pairs | self_margin], 16-B aligned pieces
    auto al = [](size_t x) { return (x + 15) & ~size_t(15); };
    const size_t o_idx = 0;
    const size_t o_mg = o_idx + al(sizeof(int32_t) * (d->n_obj_links + 1));
    const size_t o_obj = o_mg + al(sizeof(float) * (d->n_obj_links + 1));
    const size_t o_pr = o_obj + al(sizeof(DevObj) * (d->n_objects + 1));
    const size_t o_sp = o_pr + al(sizeof(DevPrim) * (d->n_prims + 1));
    const size_t o_sm = o_sp + al(sizeof(int32_t) * 2 * (d->n_self_pairs + 1));
    const size_t o_sph = o_sm + al(sizeof(float) * (d->n_self_pairs + 1));
    const size_t o_sel = o_sph + al(sizeof(float4) * (d->n_prims + 1));
    const size_t o_box = o_sel + al(sizeof(float4) * (d->n_prims + 1));
    const size_t o_pair = o_box + al(sizeof(int32_t) * (d->n_objects + 1));
    const size_t o_vsrc = o_pair + al(sizeof(float) * 8 * ((d->n_prims + 2) / 2));
    const size_t o_vw = o_vsrc + al(sizeof(int32_t) * 2 * (d->n_virtual + 1));
    const size_t total = o_vw + al(sizeof(float) * 2 * (d->n_virtual + 1));
    std::vector<char> blob(total, 0);
    std::vector<float4> spheres;
    if (d->n_obj_links) {
        std::memcpy(blob.data() + o_idx, d->obj_link_idx, sizeof(int32_t) * d->n_obj_links);
        std::memcpy(blob.data() + o_mg, d->obj_link_margin, sizeof(float) * d->n_obj_links);
    }
    DevObj* objs = reinterpret_cast<DevObj*>(blob.data() + o_obj);
    for (int o = 0; o < d->n_objects; ++o) {
        const TrkObject& ob = d->objects[o];
        std::memcpy(objs[o].pos, ob.pos, sizeof(float) * 3);
        std::memcpy(objs[o].R, ob.R, sizeof(float) * 9);
        objs[o].prim_begin = ob.prim_begin; objs[o].prim_end = ob.prim_end; objs[o].is_grid = ob.is_grid ? 1 : 0;
        const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        objs[o].identity = std::memcmp(ob.R, I, sizeof(I)) == 0 ? TRK_OBJ_IDENTITY : 0;
        if (!ob.is_grid)
            for (int p = ob.prim_begin; p < ob.prim_end; ++p) {
                const TrkPrimitive& pr = d->prims[p];
                if (pr.type != TRK_PRIM_SPHERE) { objs[o].identity |= TRK_OBJ_NONSPHERE; continue; }
                // world-frame centre: R c + pos (double accumulate, rounded once)
                float4 sp;
                double cw[3];
                for (int r = 0; r < 3; ++r)
                    cw[r] = (double)ob.R[3 * r] * pr.center[0] + (double)ob.R[3 * r + 1] * pr.center[1] +
                            (double)ob.R[3 * r + 2] * pr.center[2] + (double)ob.pos[r];
                sp.x = (float)cw[0]; sp.y = (float)cw[1]; sp.z = (float)cw[2]; sp.w = pr.radius;
                spheres.push_back(sp);
            }
    }
    DevPrim* prims = reinterpret_cast<DevPrim*>(blob.data() + o_pr);
    for (int p = 0; p < d->n_prims; ++p) {
        const TrkPrimitive& pr = d->prims[p];
        prims[p].type = pr.type;
        prims[p].cx = pr.center[0]; prims[p].cy = pr.center[1]; prims[p].cz = pr.center[2];
        prims[p].hx = pr.half[0]; prims[p].hy = pr.half[1]; prims[p].hz = pr.half[2];
        prims[p].r = pr.type == TRK_PRIM_SHARP_BOX ? 0.0f : pr.radius;
    }
    int32_t* sp = reinterpret_cast<int32_t*>(blob.data() + o_sp);
    for (int p = 0; p < 2 * d->n_self_pairs; ++p) sp[p] = d->self_link_idx[d->self_pairs[p]];
    bool self_single = false;
    for (int p = 0; p < d->n_self_pairs; ++p) self_single = self_single || sp[2 * p] == sp[2 * p + 1];
    if (d->n_virtual) {
        std::memcpy(blob.data() + o_vsrc, d->virtual_src, sizeof(int32_t) * 2 * d->n_virtual);
        std::memcpy(blob.data() + o_vw, d->virtual_w, sizeof(float) * 2 * d->n_virtual);
    }
    if (d->n_self_pairs) std::memcpy(blob.data() + o_sm, d->self_margin, sizeof(float) * d->n_self_pairs);
    const size_t n_real_spheres = spheres.size();
    // An odd table gets a copy of its last sphere appended: the ranking loop then works on whole PAIRS, and whichever index
    // of the duplicate wins addresses the same centre (n_spheres stays the real count for every other path).
    if (spheres.size() & 1) spheres.push_back(spheres.back());
    if (!spheres.empty()) std::memcpy(blob.data() + o_sph, spheres.data(), sizeof(float4) * spheres.size());
    {
        float4* sel = reinterpret_cast<float4*>(blob.data() + o_sel);
        for (size_t k = 0; k < spheres.size(); ++k) {
            const float4& sp = spheres[k];
            sel[k].x = -2.0f * sp.x; sel[k].y = -2.0f * sp.y; sel[k].z = -2.0f * sp.z;
            sel[k].w = (float)((double)sp.x * sp.x + (double)sp.y * sp.y + (double)sp.z * sp.z);
        }
        // the same rows, two spheres (S, T) interleaved per record: [Sx Tx | Sy Ty | Sz Tz | Sw Tw] -- the operand layout of
        // v_pk_fma_f32 with one packed lane per sphere
        float* pair = reinterpret_cast<float*>(blob.data() + o_pair);
        for (size_t j = 0; j + 1 < spheres.size(); j += 2) {
            const float4 &S = sel[j], &T = sel[j + 1];
            float* r = pair + 4 * j;
            r[0] = S.x; r[1] = T.x; r[2] = S.y; r[3] = T.y; r[4] = S.z; r[5] = T.z; r[6] = S.w; r[7] = T.w;
        }
    }
    int n_box = 0;
    {
        int32_t* box = reinterpret_cast<int32_t*>(blob.data() + o_box);
        for (int o = 0; o < d->n_objects; ++o)
            if (!d->objects[o].is_grid && (objs[o].identity & TRK_OBJ_NONSPHERE)) box[n_box++] = o;
    }
    bool uniform_r = !spheres.empty();
    for (const float4& sp : spheres) uniform_r = uniform_r && sp.w == spheres[0].w;
    const int n_sphere_pairs = (int)(spheres.size() / 2);

    TrkCostModel* cm = new (std::nothrow) TrkCostModel();
    if (!cm) return fail(TRK_ERR_HIP, "out of host memory");
    hipError_t e = hipMalloc(&cm->d_blob, total);
    if (e == hipSuccess) e = hipMemcpy(cm->d_blob, blob.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (cm->d_blob) (void)hipFree(cm->d_blob);
        delete cm;
        return hip_fail(e, "trk_cost_model_create: device allocation/copy");
    }
    DevCostHdr& h = cm->hdr;
    std::memset(&h, 0, sizeof(h));
    char* base = static_cast<char*>(cm->d_blob);
    h.n_links_in = Lin;
    h.n_obj_links = d->n_obj_links; h.n_objects = d->n_objects; h.has_grid = n_grid; h.has_ws = d->has_ws ? 1 : 0;
    h.n_self_links = d->n_self_links; h.n_self_pairs = d->n_self_pairs;
    h.ee_link = d->ee_link; h.ee_square = d->ee_square ? 1 : 0; h.ee_w_pos = d->ee_w_pos; h.ee_w_rot = d->ee_w_rot;
    std::memcpy(h.ws_min, d->ws_min, sizeof(float) * 3);
    std::memcpy(h.ws_max, d->ws_max, sizeof(float) * 3);
    for (int k = 0; k < 3; ++k) { h.ws_c[k] = 0.5f * (d->ws_min[k] + d->ws_max[k]); h.ws_h[k] = 0.5f * (d->ws_max[k] - d->ws_min[k]); }
    std::memcpy(h.ee_target, d->ee_target, sizeof(float) * 16);
    h.ee2_link = d->ee2_link;
    std::memcpy(h.ee2_target, d->ee2_target, sizeof(float) * 16);
    h.obj_link_idx = reinterpret_cast<const int32_t*>(base + o_idx);
    h.obj_link_margin = reinterpret_cast<const float*>(base + o_mg);
    h.objects = reinterpret_cast<const DevObj*>(base + o_obj);
    h.prims = reinterpret_cast<const DevPrim*>(base + o_pr);
    h.n_prims = d->n_prims; h._pad_prims = 0;
    h.self_pairs = reinterpret_cast<const int32_t*>(base + o_sp);
    h.self_margin = reinterpret_cast<const float*>(base + o_sm);
    cm->obj_link_idx.assign(d->obj_link_idx, d->obj_link_idx + d->n_obj_links);
    cm->self_pairs.assign(sp, sp + 2 * d->n_self_pairs);
    if (d->n_virtual) {
        cm->virtual_src.assign(d->virtual_src, d->virtual_src + 2 * d->n_virtual);
        cm->virtual_w.assign(d->virtual_w, d->virtual_w + 2 * d->n_virtual);
    }
    h.spheres = reinterpret_cast<const float4*>(base + o_sph);
    h.spheres_sel = reinterpret_cast<const float4*>(base + o_sel);
    h.box_objects = reinterpret_cast<const int32_t*>(base + o_box);
    h.n_box_objects = n_box;
    h.n_spheres = (int32_t)n_real_spheres;
    h.sphere_pairs = reinterpret_cast<const float*>(base + o_pair);
    h.n_sphere_pairs = n_sphere_pairs;
    h.clamp_fields = d->clamp_fields & 7;
    h.n_virtual = d->n_virtual; h.self_single = self_single ? 1 : 0;
    h.virtual_src = reinterpret_cast<const int32_t*>(base + o_vsrc);
    h.virtual_w = reinterpret_cast<const float*>(base + o_vw);
    h.spheres_uniform_r = uniform_r ? 1 : 0;
    h.sphere_r = spheres.empty() ? 0.0f : spheres[0].w;
    if (n_grid) {
        // the kernels gather one 16-byte record per point: pack the caller's arrays once (a snapshot, like every other table)
        const int64_t n_cells = (int64_t)d->grid.dims[0] * d->grid.dims[1] * d->grid.dims[2];
        const int nb0 = (d->grid.dims[0] + 3) / 4, nb1 = (d->grid.dims[1] + 3) / 4, nb2 = (d->grid.dims[2] + 3) / 4;
        const int64_t n_rec = (int64_t)nb0 * nb1 * nb2 * 64;           // whole 4 x 4 x 4 bricks (grid_record)
        h.grid.nb1 = nb1; h.grid.nb2 = nb2;
        e = n_cells > 0 ? hipMalloc(&cm->d_cells, sizeof(float4) * (size_t)n_rec) : hipErrorInvalidValue;
        if (e == hipSuccess) {
            trk_launch_grid_pack(d->grid.sdf, d->grid.grad, d->grid.dims, nb1, nb2, n_rec, cm->d_cells, nullptr);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        }
        if (e != hipSuccess) {
            if (cm->d_cells) (void)hipFree(cm->d_cells);
            (void)hipFree(cm->d_blob);
            delete cm;
            return hip_fail(e, "trk_cost_model_create: packing the voxel grid");
        }
        h.grid.cells = cm->d_cells;
        for (int k = 0; k < 3; ++k) {
            h.grid.dims[k] = d->grid.dims[k]; h.grid.lim_min[k] = d->grid.lim_min[k];
            h.grid.map_dim[k] = d->grid.map_dim[k]; h.grid.fdims[k] = (float)d->grid.dims[k];
        }
    }
    live_add(cm, 2);
    *out = cm;
    return TRK_OK;
}

void trk_cost_model_destroy(TrkCostModel* cm) {
    if (!cm) return;
    live_del(cm);
    if (cm->d_blob) (void)hipFree(cm->d_blob);
    if (cm->d_cells) (void)hipFree(cm->d_cells);
    delete cm;
}

int trk_cost_model_enable_specialized(TrkCostModel* cm, int32_t on) {
    if (!cm) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_enable_specialized: null cost model");
    cm->spec_enabled = on != 0;
    return TRK_OK;
}

int trk_cost_model_set_ee_target(TrkCostModel* cm, const float* H16) {
    if (!cm || !H16) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_set_ee_target: null argument");
    std::memcpy(cm->hdr.ee_target, H16, sizeof(float) * 16);
    return TRK_OK;
}

int trk_cost_model_set_ee2_target(TrkCostModel* cm, const float* H16) {
    if (!cm || !H16) return fail(TRK_ERR_INVALID_ARG, "trk_cost_model_set_ee2_target: null argument");
    std::memcpy(cm->hdr.ee2_target, H16, sizeof(float) * 16);
    return TRK_OK;
}

int trk_cost_fields(const TrkCostModel* cm, int32_t fields, const float* link_pos, int64_t n, const float* gcost,
                    float* cost, float* g_link_pos, trk_stream_t stream) {
    if (!cm) return fail(TRK_ERR_INVALID_ARG, "trk_cost_fields: null cost model");
    if (n < 0 || (n > 0 && (!link_pos || !cost)) || (fields & ~7) || !fields) return fail(TRK_ERR_INVALID_ARG, "trk_cost_fields: bad argument");
    if (n == 0) return TRK_OK;
    {
        const TrkRolloutWeights w = effective_weights(cm, TrkRolloutWeights{(fields & TRK_FIELD_SELF) ? 1.0f : 0.0f,
                                                                            (fields & TRK_FIELD_OBJECTS) ? 1.0f : 0.0f,
                                                                            (fields & TRK_FIELD_WS) ? 1.0f : 0.0f, 0.0f});
        if (const SpecEntry* e = fields_spec_for(cm, &w)) {      // the fused kernel's objective code on the caller's positions
            SpecArgs a{};
            a.C = cm->hdr; a.w = w;
            a.n = n; a.fld_pos = link_pos; a.fld_gcost = gcost; a.fld_g = g_link_pos; a.cost = cost;
            e->launch_fields(e, a, 1, (hipStream_t)stream);
            TRK_HIP(last_launch_error());
            return TRK_OK;
        }
    }
    trk_launch_cost_fields(cm->hdr, fields, link_pos, n, gcost, cost, g_link_pos, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_collision_fields(const TrkCostModel* cm, int32_t fields, const float* link_pos, int64_t n, float margin_override,
                         uint8_t* in_collision, trk_stream_t stream) {
    if (!cm) return fail(TRK_ERR_INVALID_ARG, "trk_collision_fields: null cost model");
    if (n < 0 || (n > 0 && (!link_pos || !in_collision)) || (fields & ~7) || !fields) return fail(TRK_ERR_INVALID_ARG, "trk_collision_fields: bad argument");
    if (n == 0) return TRK_OK;
    const int use_default = std::isnan(margin_override) ? 1 : 0;
    {
        TrkRolloutWeights w{(fields & TRK_FIELD_SELF) ? 1.0f : 0.0f, (fields & TRK_FIELD_OBJECTS) ? 1.0f : 0.0f,
                            (fields & TRK_FIELD_WS) ? 1.0f : 0.0f, 0.0f};
        if (const SpecEntry* e = fields_spec_for(cm, &w)) {      // the fused boolean kernel's tests on the caller's positions
            SpecArgs a{};
            a.C = cm->hdr; a.w = w;
            a.n = n; a.fld_pos = link_pos; a.coll_out = in_collision; a.coll_fields = fields;
            a.coll_use_default = use_default; a.coll_margin = use_default ? 0.0f : margin_override;
            e->launch_fields(e, a, 1, (hipStream_t)stream);
            TRK_HIP(last_launch_error());
            return TRK_OK;
        }
    }
    trk_launch_collision_fields(cm->hdr, fields, link_pos, n, use_default ? 0.0f : margin_override, use_default, in_collision, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_ee_cost(const TrkCostModel* cm, const float* H_ee, int64_t n, int64_t stride, const float* target, int32_t per_sample,
                const float* gcost, float* cost, float* gH, int64_t g_stride, trk_stream_t stream) {
    if (!cm) return fail(TRK_ERR_INVALID_ARG, "trk_ee_cost: null cost model");
    if (n < 0 || stride < 16 || (stride & 3) || (gH && (g_stride < 16 || (g_stride & 3))) || (n > 0 && (!H_ee || !cost)))
        return fail(TRK_ERR_INVALID_ARG, "trk_ee_cost: bad argument (strides must be multiples of 4 floats, >= 16)");
    if (n == 0) return TRK_OK;
    trk_launch_ee_cost(cm->hdr, H_ee, n, stride, target, per_sample, gcost, cost, gH, g_stride, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

static int rollout_impl(const char* who, const TrkModel* m, const TrkCostModel* cm, const TrkRolloutWeights* w, int io_f16, float grad_scale,
                        const void* q, int64_t batch, int32_t horizon, void* link_pos_out, float* cost, void* gq,
                        float* cost_sum, trk_stream_t stream) {
    int rc = check_model(m, who);
    if (rc) return rc;
    if (!cm || !w) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": null argument");
    if (batch < 0 || horizon < 1) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": bad batch/horizon");
    if (cm->hdr.n_links_in != m->hdr.n_links) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": cost model n_links_in != model n_links");
    const int64_t n = batch * horizon;
    if (n > 0 && (!q || !cost || !gq)) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": null q/cost/gq");
    if (n == 0) return TRK_OK;
    const TrkRolloutWeights we = effective_weights(cm, *w);
    w = &we;
    if (m->spec_enabled) {
        // a generated kernel has the robot's collision-link sets baked in: use the unit whose sets equal the cost model's
        if (const SpecEntry* e = model_spec_for(m, cm, w)) {
            SpecArgs a{};
            a.C = cm->hdr; a.w = *w;
            std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
            std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
            a.q = q; a.n = n; a.link_pos = link_pos_out; a.cost = cost; a.gq = gq; a.cost_sum = cost_sum;
            a.stamps = g_stamps; a.io_f16 = io_f16; a.grad_scale = grad_scale;
            e->launch(e, a, base_is_identity(m), (hipStream_t)stream);
            TRK_HIP(last_launch_error());
            g_last_dispatch = TRK_DISPATCH_GENERATED;
            return TRK_OK;
        }
    }
    if ((rc = strict_refusal(who, m)) != TRK_OK) return rc;
    g_last_dispatch = TRK_DISPATCH_TABLE;
    if (trk_lds_rollout(m->hdr, m->hdr.n_links + cm->hdr.n_virtual) > kMaxLds) return fail(TRK_ERR_UNSUPPORTED, std::string(who) + ": position tiles (links + interpolated points) exceed the 160 KiB LDS");
    trk_launch_rollout_generic(m->hdr, m->d_links, m->d_fin, nullptr, cm->hdr, *w, io_f16, grad_scale, q, n, link_pos_out, cost, gq, cost_sum, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_last_dispatch(void) { return g_last_dispatch; }
int trk_set_strict_specialized(int on) {
    const int prev = strict_specialized() ? 1 : 0;
    g_strict = on ? 1 : 0;
    return prev;
}

int trk_rollout_is_specialized(const TrkModel* m, const TrkCostModel* cm, const TrkRolloutWeights* w) {
    if (!m || !cm || !w || !m->spec_enabled || cm->hdr.n_links_in != m->hdr.n_links) return 0;
    const TrkRolloutWeights we = effective_weights(cm, *w);
    return model_spec_for(m, cm, &we) ? 1 : 0;
}

int trk_rollout_points_is_specialized(const TrkPointSet* ps, const TrkCostModel* cm, const TrkRolloutWeights* w) {
    if (!ps || !ps->model || !cm || !w || !ps->model->spec_enabled || cm->hdr.n_links_in != ps->dev.n_points) return 0;
    const TrkRolloutWeights we = effective_weights(cm, *w);
    return points_spec_for(ps, cm, &we) ? 1 : 0;
}

int trk_rollout_collision(const TrkModel* m, const TrkCostModel* cm, int32_t fields, const float* q, int64_t batch, int32_t horizon,
                          float margin_override, uint8_t* in_collision, float* link_pos_ws, trk_stream_t stream) {
    int rc = check_model(m, "trk_rollout_collision");
    if (rc) return rc;
    if (!cm) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_collision: null cost model");
    if (batch < 0 || horizon < 1 || (fields & ~7) || !fields) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_collision: bad batch / horizon / fields");
    if (cm->hdr.n_links_in != m->hdr.n_links) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_collision: cost model n_links_in != model n_links");
    const int64_t n = batch * horizon;
    if (n > 0 && (!q || !in_collision)) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_collision: null q / in_collision");
    if (n == 0) return TRK_OK;
    const int use_default = std::isnan(margin_override) ? 1 : 0;
    fields = effective_fields(cm, fields);
    if (!fields) {                              // nothing to test: nobody collides
        TRK_HIP(hipMemsetAsync(in_collision, 0, (size_t)n, (hipStream_t)stream));
        g_last_dispatch = TRK_DISPATCH_NONE;
        return TRK_OK;
    }
    if (m->spec_enabled) {
        // the unit's baked link sets must equal the cost model's for every field that is asked for
        TrkRolloutWeights w{};
        w.w_self = (fields & TRK_FIELD_SELF) ? 1.0f : 0.0f;
        w.w_obj = (fields & (TRK_FIELD_OBJECTS | TRK_FIELD_WS)) ? 1.0f : 0.0f;
        w.w_ws = (fields & TRK_FIELD_WS) ? 1.0f : 0.0f;
        const SpecEntry* e = model_spec_for(m, cm, &w);
        if (e && e->launch_coll) {
            SpecArgs a{};
            a.C = cm->hdr; a.w = w;
            std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
            std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
            a.q = q; a.n = n;
            a.coll_out = in_collision; a.coll_fields = fields; a.coll_use_default = use_default;
            a.coll_margin = use_default ? 0.0f : margin_override;
            e->launch_coll(e, a, base_is_identity(m), (hipStream_t)stream);
            TRK_HIP(last_launch_error());
            g_last_dispatch = TRK_DISPATCH_GENERATED;
            return TRK_OK;
        }
    }
    if ((rc = strict_refusal("trk_rollout_collision", m)) != TRK_OK) return rc;
    g_last_dispatch = TRK_DISPATCH_TABLE;
    // no generated unit serves this model / cost model: table-driven FK into the caller's scratch, then the field kernel
    if (!link_pos_ws) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_collision: no generated kernel for this model and no link_pos_ws scratch given");
    rc = trk_fk_positions(m, q, n, nullptr, 0, link_pos_ws, stream);
    if (rc) return rc;
    trk_launch_collision_fields(cm->hdr, fields, link_pos_ws, n, use_default ? 0.0f : margin_override, use_default, in_collision, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

static int rollout_collision_via_impl(const TrkModel* m, const TrkCostModel* cm, int32_t fields, const float* x, int64_t n_traj,
                                      int32_t horizon, int32_t state_dim, int32_t n_interp, const float* alpha, const float* beta,
                                      float margin_override, uint8_t* in_collision, uint8_t* traj_flags, const float* q_min, const float* q_max,
                                      trk_stream_t stream);

int trk_rollout_collision_via(const TrkModel* m, const TrkCostModel* cm, int32_t fields, const float* x, int64_t n_traj,
                              int32_t horizon, int32_t state_dim, int32_t n_interp, const float* alpha, const float* beta,
                              float margin_override, uint8_t* in_collision, trk_stream_t stream) {
    return rollout_collision_via_impl(m, cm, fields, x, n_traj, horizon, state_dim, n_interp, alpha, beta, margin_override, in_collision,
                                      nullptr, nullptr, nullptr, stream);
}

int64_t trk_via_partial_flags_bytes(int64_t n_traj, int32_t horizon, int32_t n_interp) {
    if (n_traj < 0 || horizon < 2 || n_interp < 1) return 0;
    const int64_t hi = (int64_t)(horizon - 1) * n_interp;
    return n_traj * (hi / TRK_WAVE + 2);          // one byte per trajectory and wavefront that can hold samples of it
}

int trk_rollout_collision_via_flags(const TrkModel* m, const TrkCostModel* cm, int32_t fields, const float* x, int64_t n_traj,
                                    int32_t horizon, int32_t state_dim, int32_t n_interp, const float* alpha, const float* beta,
                                    float margin_override, const float* q_min, const float* q_max, uint8_t* in_collision,
                                    uint8_t* traj_flags, trk_stream_t stream) {
    if (!traj_flags || !q_min || !q_max)
        return fail(TRK_ERR_INVALID_ARG, "trk_rollout_collision_via_flags: partial_flags (trk_via_partial_flags_bytes), q_min and q_max are required");
    return rollout_collision_via_impl(m, cm, fields, x, n_traj, horizon, state_dim, n_interp, alpha, beta, margin_override, in_collision,
                                      traj_flags, q_min, q_max, stream);
}

static int rollout_collision_via_impl(const TrkModel* m, const TrkCostModel* cm, int32_t fields, const float* x, int64_t n_traj,
                                      int32_t horizon, int32_t state_dim, int32_t n_interp, const float* alpha, const float* beta,
                                      float margin_override, uint8_t* in_collision, uint8_t* traj_flags, const float* q_min, const float* q_max,
                                      trk_stream_t stream) {
    int rc = check_model(m, "trk_rollout_collision_via");
    if (rc) return rc;
    if (!cm) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_collision_via: null cost model");
    if (n_traj < 0 || horizon < 2 || n_interp < 1 || state_dim < m->hdr.n_dofs || (fields & ~7) || !fields || !alpha || !beta)
        return fail(TRK_ERR_INVALID_ARG, "trk_rollout_collision_via: bad argument");
    if (cm->hdr.n_links_in != m->hdr.n_links) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_collision_via: cost model n_links_in != model n_links");
    const int64_t hi = (int64_t)(horizon - 1) * n_interp;
    if (hi > 0x7fffffff - 64) return fail(TRK_ERR_UNSUPPORTED, "trk_rollout_collision_via: (horizon - 1) * n_interp too large");
    const int64_t n = n_traj * hi;
    if (n > 0 && (!x || !in_collision)) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_collision_via: null x / in_collision");
    if (n == 0) return TRK_OK;
    if (!m->spec_enabled) return fail(TRK_ERR_UNSUPPORTED, "trk_rollout_collision_via: generated kernels are disabled for this model");
    const int use_default = std::isnan(margin_override) ? 1 : 0;
    fields = effective_fields(cm, fields);
    // (the fused flags need the generated kernel even when no field has anything to test: the joint limits are looked at there)
    if (!fields && !traj_flags) {
        TRK_HIP(hipMemsetAsync(in_collision, 0, (size_t)n, (hipStream_t)stream));
        g_last_dispatch = TRK_DISPATCH_NONE;
        return TRK_OK;
    }
    TrkRolloutWeights w{};
    w.w_self = (fields & TRK_FIELD_SELF) ? 1.0f : 0.0f;
    w.w_obj = (fields & (TRK_FIELD_OBJECTS | TRK_FIELD_WS)) ? 1.0f : 0.0f;
    w.w_ws = (fields & TRK_FIELD_WS) ? 1.0f : 0.0f;
    const SpecEntry* e = model_spec_for(m, cm, &w);
    if (!e || !e->launch_coll)
        return fail(TRK_ERR_UNSUPPORTED, "trk_rollout_collision_via: no generated kernel serves this model / cost model "
                                         "(use trk_interpolate_via_points + trk_rollout_collision)");
    SpecArgs a{};
    a.C = cm->hdr; a.w = w;
    std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
    std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
    a.q = x; a.n = n;
    a.coll_out = in_collision; a.coll_fields = fields; a.coll_use_default = use_default;
    a.coll_margin = use_default ? 0.0f : margin_override;
    a.via_alpha = alpha; a.via_beta = beta; a.via_n = n_interp; a.via_H = horizon; a.via_S = state_dim;
    if (traj_flags) { a.via_partial = traj_flags; a.via_slots = trk_via_slots(hi); a.via_qmin = q_min; a.via_qmax = q_max; }
    e->launch_coll(e, a, base_is_identity(m), (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    g_last_dispatch = TRK_DISPATCH_GENERATED;
    return TRK_OK;
}

int trk_traj_validate(const uint8_t* waypoint_collisions, const float* x, int64_t n_traj, int32_t horizon, int32_t state_dim,
                      int32_t n_waypoints, int32_t n_dofs, const float* q_min, const float* q_max, int64_t inner,
                      uint8_t* flags, int64_t* idx, int32_t* counts, int32_t* counts_host, int32_t ticket, float* gathered,
                      trk_stream_t stream) {
    const bool have_flags = n_waypoints < 0;         // waypoint_collisions = the via-point launch's per-wavefront partial flags, -n_waypoints samples per trajectory
    if (n_traj < 0 || n_traj > 0x3fffffff || horizon < 1 || state_dim < 1 || n_dofs < 0 || n_dofs > state_dim || inner < 0 ||
        !counts || (!have_flags && n_dofs > 0 && (!q_min || !q_max)) ||
        (n_traj > 0 && (!x || !flags || !idx || (n_waypoints != 0 && !waypoint_collisions))))
        return fail(TRK_ERR_INVALID_ARG, "trk_traj_validate: bad argument");      // n_traj == 0: only the (zero) counters are written
    if (inner > 0 && n_traj % inner) return fail(TRK_ERR_INVALID_ARG, "trk_traj_validate: n_traj is not a multiple of the inner batch");
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_traj_validate(waypoint_collisions, x, n_traj, horizon, state_dim, n_waypoints, n_dofs, q_min, q_max, inner, flags,
                             idx, counts, counts_host, ticket, gathered, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_rollout_jacobian_cost_grad(const TrkModel* m, const TrkCostModel* cm, const TrkRolloutWeights* w, const float* q, int64_t batch,
                                   int32_t horizon, int32_t link, float* link_pos_out, float* cost, float* gq, float* cost_sum, float* pos,
                                   float* quat, float* lin_jac, float* ang_jac, trk_stream_t stream) {
    const char* who = "trk_rollout_jacobian_cost_grad";
    int rc = check_model(m, who);
    if (rc) return rc;
    if (!cm || !w) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": null argument");
    if (batch < 0 || horizon < 1 || link < 0 || link >= m->hdr.n_links) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": bad batch / horizon / link");
    if (cm->hdr.n_links_in != m->hdr.n_links) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": cost model n_links_in != model n_links");
    const int64_t n = batch * horizon;
    if (n > 0 && (!q || !cost || !gq || !pos || !quat || !lin_jac || !ang_jac)) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": null q / cost / gq / Jacobian output");
    if (n == 0) return TRK_OK;
    const TrkRolloutWeights we = effective_weights(cm, *w);
    if (m->spec_enabled) {
        const SpecEntry* e = model_spec_for(m, cm, &we);
        if (e && e->launch_rjac) {
            SpecArgs a{};
            a.C = cm->hdr; a.w = we;
            std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
            std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
            a.q = q; a.n = n; a.link_pos = link_pos_out; a.cost = cost; a.gq = gq; a.cost_sum = cost_sum;
            a.stamps = g_stamps; a.io_f16 = 0; a.grad_scale = 1.0f;
            a.jac_link = link; a.jac_joint_idx = m->joint_list_idx[link];
            a.jac_pos = pos; a.jac_quat = quat; a.jac_lin = lin_jac; a.jac_ang = ang_jac;
            // q in; positions, cost, gradient, pos, quat, lin_jac, ang_jac out: beyond the Infinity Cache the Jacobian tiles stream
            a.jac_stream = spec_stream_bytes((double)n * (8.0 * m->hdr.n_dofs + 4.0 + (link_pos_out ? 12.0 * m->hdr.n_links : 0.0) + 28.0 +
                                                          24.0 * m->hdr.n_dofs)) ? 1 : 0;
            if (e->launch_rjac(e, a, base_is_identity(m), (hipStream_t)stream) == 0) {
                TRK_HIP(last_launch_error());
                g_last_dispatch = TRK_DISPATCH_GENERATED;
                return TRK_OK;
            }
        }
    }
    // the two-launch form: the fused rollout, then the Jacobian kernel (a second walk of the chain)
    rc = rollout_impl(who, m, cm, w, 0, 1.0f, q, batch, horizon, link_pos_out, cost, gq, cost_sum, stream);
    if (rc) return rc;
    const int d = g_last_dispatch;
    rc = trk_fk_jacobian(m, q, nullptr, n, link, pos, quat, lin_jac, ang_jac, nullptr, nullptr, stream);
    g_last_dispatch = d == TRK_DISPATCH_GENERATED ? TRK_DISPATCH_GENERATED_PLUS_PRIOR : d;
    return rc;
}

int trk_rollout_cost_grad(const TrkModel* m, const TrkCostModel* cm, const TrkRolloutWeights* w, const float* q,
                          int64_t batch, int32_t horizon, float* link_pos_out, float* cost, float* gq, float* cost_sum,
                          trk_stream_t stream) {
    return rollout_impl("trk_rollout_cost_grad", m, cm, w, 0, 1.0f, q, batch, horizon, link_pos_out, cost, gq, cost_sum, stream);
}

int trk_rollout_cost_grad_f16(const TrkModel* m, const TrkCostModel* cm, const TrkRolloutWeights* w, const void* q_f16,
                              int64_t batch, int32_t horizon, void* link_pos_out_f16, float* cost, void* gq,
                              int32_t grad_dtype, float grad_scale, float* cost_sum, trk_stream_t stream) {
    if ((grad_dtype != TRK_F32 && grad_dtype != TRK_F16) || !(grad_scale > 0.0f) || !std::isfinite(grad_scale))
        return fail(TRK_ERR_INVALID_ARG, "trk_rollout_cost_grad_f16: grad_dtype must be TRK_F32 / TRK_F16, grad_scale finite and > 0");
    return rollout_impl("trk_rollout_cost_grad_f16", m, cm, w, grad_dtype == TRK_F16 ? 1 : 2, grad_scale, q_f16, batch, horizon,
                        link_pos_out_f16, cost, gq, cost_sum, stream);
}

int trk_rollout_gp_cost_grad(const TrkModel* m, const TrkCostModel* cm, const TrkRolloutWeights* w, const TrkGpPrior* gp,
                             const void* q, const void* qd, int64_t batch, int32_t horizon, int32_t io_dtype, void* link_pos_out,
                             float* cost, void* gq, void* gqd, int32_t grad_dtype, float grad_scale, float* cost_sum, trk_stream_t stream) {
    const char* who = "trk_rollout_gp_cost_grad";
    int rc = check_model(m, who);
    if (rc) return rc;
    if (!cm || !w || !gp) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": null argument");
    if (batch < 0 || horizon < 1) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": bad batch/horizon");
    if ((io_dtype != TRK_F32 && io_dtype != TRK_F16) || (grad_dtype != TRK_F32 && grad_dtype != TRK_F16) ||
        (io_dtype == TRK_F32 && (grad_dtype != TRK_F32 || grad_scale != 1.0f)) || !(grad_scale > 0.0f) || !std::isfinite(grad_scale))
        return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": io_dtype / grad_dtype must be TRK_F32 / TRK_F16 (fp32 trajectories: fp32 gradient, "
                                         "grad_scale 1), grad_scale finite and > 0");
    if (!(gp->dt > 0.0f) || !(gp->sigma > 0.0f)) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": the prior needs dt > 0 and sigma > 0");
    if (cm->hdr.n_links_in != m->hdr.n_links) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": cost model n_links_in != model n_links");
    const int64_t n = batch * horizon;
    if (n > 0 && (!q || !qd || !cost || !gq || !gqd)) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": null q/qd/cost/gq/gqd");
    if (n == 0) return TRK_OK;
    const int io_mode = io_dtype == TRK_F32 ? 0 : (grad_dtype == TRK_F16 ? 1 : 2);
    const TrkRolloutWeights we = effective_weights(cm, *w);
    if (m->spec_enabled) {
        const SpecEntry* e = model_spec_for(m, cm, &we);
        if (e && e->launch_gp) {
            SpecArgs a{};
            a.C = cm->hdr; a.w = we;
            std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
            std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
            a.q = q; a.n = n; a.link_pos = link_pos_out; a.cost = cost; a.gq = gq; a.cost_sum = cost_sum;
            a.stamps = g_stamps; a.io_f16 = io_mode; a.grad_scale = grad_scale;
            const float s2 = 1.0f / (gp->sigma * gp->sigma);
            a.qd = qd; a.gqd = gqd; a.gp_dt = gp->dt; a.gp_w = gp->weight; a.gp_H = horizon;
            a.gp_a = 12.0f * s2 / (gp->dt * gp->dt * gp->dt); a.gp_b = -6.0f * s2 / (gp->dt * gp->dt); a.gp_c = 4.0f * s2 / gp->dt;
            if (e->launch_gp(e, a, base_is_identity(m), (hipStream_t)stream) == 0) {
                TRK_HIP(last_launch_error());
                g_last_dispatch = TRK_DISPATCH_GENERATED;
                return TRK_OK;
            }
        }
    }
    // the two-launch form: the rollout, then the prior accumulated into its gradient, its factor costs into the per-sample costs
    rc = rollout_impl(who, m, cm, w, io_mode, grad_scale, q, batch, horizon, link_pos_out, cost, gq, nullptr, stream);
    if (rc) return rc;
    if (g_last_dispatch == TRK_DISPATCH_GENERATED) g_last_dispatch = TRK_DISPATCH_GENERATED_PLUS_PRIOR;     // generated rollout, the prior as launches of its own
    rc = ensure_init();
    if (rc) return rc;
    if (batch > 0x7fffffff) return fail(TRK_ERR_INVALID_ARG, std::string(who) + ": batch too large");
    // gqd is overwritten, gq accumulated: two calls of the prior kernel would read q twice, so gqd is zeroed and both are accumulated
    TRK_HIP(hipMemsetAsync(gqd, 0, (size_t)n * m->hdr.n_dofs * (grad_dtype == TRK_F16 ? 2 : 4), (hipStream_t)stream));
    // (the prior kernel's per-trajectory cost is not an output of this entry point: nullptr)
    if (trk_launch_gp_prior(io_dtype == TRK_F16, grad_dtype == TRK_F16, io_dtype == TRK_F16 ? grad_scale : 1.0f, q, qd, batch, horizon,
                            m->hdr.n_dofs, gp->dt, gp->sigma, gp->weight, nullptr, gq, gqd, 1, (hipStream_t)stream))
        return fail(TRK_ERR_UNSUPPORTED, std::string(who) + ": one trajectory (horizon x dof x 2 floats) must fit the 160 KiB LDS");
    trk_launch_gp_sample_cost(io_dtype == TRK_F16, q, qd, n, horizon, m->hdr.n_dofs, gp->dt, gp->sigma, gp->weight, cost, cost_sum,
                              (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_rollout_points_cost_grad(const TrkModel* m, const TrkPointSet* ps, const TrkCostModel* cm, const TrkRolloutWeights* w,
                                 const float* q, int64_t batch, int32_t horizon, float* point_pos_out, float* cost, float* gq,
                                 float* cost_sum, trk_stream_t stream) {
    int rc = check_model(m, "trk_rollout_points_cost_grad");
    if (rc) return rc;
    if (!cm || !w) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_points_cost_grad: null argument");
    if (!ps || ps->model != m) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_points_cost_grad: point set does not belong to this model");
    if (batch < 0 || horizon < 1) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_points_cost_grad: bad batch/horizon");
    if (cm->hdr.n_links_in != ps->dev.n_points) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_points_cost_grad: cost model n_links_in != number of points");
    if (cm->hdr.ee_link >= m->hdr.n_links || cm->hdr.ee2_link >= m->hdr.n_links)
        return fail(TRK_ERR_INVALID_ARG, "trk_rollout_points_cost_grad: ee_link is not a link of the model");
    if (trk_lds_rollout(m->hdr, ps->dev.n_points + cm->hdr.n_virtual) > kMaxLds) return fail(TRK_ERR_UNSUPPORTED, "trk_rollout_points_cost_grad: point tiles exceed the 160 KiB LDS");
    const int64_t n = batch * horizon;
    if (n > 0 && (!q || !cost || !gq)) return fail(TRK_ERR_INVALID_ARG, "trk_rollout_points_cost_grad: null q/cost/gq");
    if (n == 0) return TRK_OK;
    const TrkRolloutWeights we = effective_weights(cm, *w);
    w = &we;
    if (m->spec_enabled && (reinterpret_cast<uintptr_t>(point_pos_out) & 15) == 0) {
        // generated kernel with this point set baked in whose cost columns equal the cost model's
        if (const SpecEntry* e = points_spec_for(ps, cm, w)) {
            SpecArgs a{};
            a.C = cm->hdr; a.w = *w;
            std::memcpy(a.base_R, m->hdr.base_R, sizeof(a.base_R));
            std::memcpy(a.base_t, m->hdr.base_t, sizeof(a.base_t));
            a.q = q; a.n = n; a.link_pos = point_pos_out; a.cost = cost; a.gq = gq; a.cost_sum = cost_sum;
            a.stamps = nullptr; a.io_f16 = 0;
            e->launch(e, a, base_is_identity(m), (hipStream_t)stream);
            TRK_HIP(last_launch_error());
            g_last_dispatch = TRK_DISPATCH_GENERATED;
            return TRK_OK;
        }
    }
    if (strict_specialized() && m->spec_enabled && points_spec(ps))
        return fail(TRK_ERR_UNSUPPORTED, "trk_rollout_points_cost_grad: strict mode: the point set has generated kernels but none bakes this cost model's columns "
                                         "for the non-zero weights (or point_pos_out is not 16-byte aligned) -- the call would take the table-driven kernel");
    g_last_dispatch = TRK_DISPATCH_TABLE;
    trk_launch_rollout_generic(m->hdr, m->d_links, m->d_fin, &ps->dev, cm->hdr, *w, 0, 1.0f, q, n, point_pos_out, cost, gq, cost_sum, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_interpolate_via_points(const float* x, int64_t n_traj, int32_t horizon, int32_t dim, int32_t n_interp,
                               const float* alpha, const float* beta, float* out, trk_stream_t stream) {
    if (n_traj < 0 || horizon < 2 || dim < 1 || n_interp < 1 || !alpha || !beta || (n_traj > 0 && (!x || !out)))
        return fail(TRK_ERR_INVALID_ARG, "trk_interpolate_via_points: bad argument");
    if (n_traj == 0) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_interpolate(x, n_traj, horizon, dim, n_interp, alpha, beta, out, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_jtj(const float* lin_jac, const float* ang_jac, const float* residual, int64_t n, int32_t dof, int32_t use_mfma,
            float* JtJ, float* Jtr, const float* damping, int32_t damping_stride, float* dq, trk_stream_t stream) {
    if (n < 0 || dof < 1 || dof > TRK_MAX_DOFS || (n > 0 && (!lin_jac || !ang_jac || !JtJ)) || ((Jtr || dq) && !residual) ||
        (damping && damping_stride != 0 && damping_stride != 1))
        return fail(TRK_ERR_INVALID_ARG, "trk_jtj: bad argument");
    if (use_mfma && dof > 8) return fail(TRK_ERR_UNSUPPORTED, "trk_jtj: the MFMA kernel tiles an 8 x 8 matrix per sample (dof <= 8)");
    if (n == 0) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    if (trk_launch_jtj(use_mfma != 0, lin_jac, ang_jac, residual, n, dof, JtJ, Jtr, damping, damping_stride, dq, (hipStream_t)stream))
        return fail(TRK_ERR_UNSUPPORTED, "trk_jtj: tiles exceed the 160 KiB LDS");
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_scale_rows(const void* g, const float* scale, int32_t scale_stride, int64_t n, int32_t dim, int32_t io_dtype, void* out,
                   trk_stream_t stream) {
    if (n < 0 || dim < 1 || (scale_stride != 0 && scale_stride != 1) || (io_dtype != TRK_F32 && io_dtype != TRK_F16) ||
        (n > 0 && (!g || !scale || !out)))
        return fail(TRK_ERR_INVALID_ARG, "trk_scale_rows: bad argument");
    if (n == 0) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_scale_rows(io_dtype == TRK_F16, g, scale, scale_stride, n, dim, out, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_interpolate_columns(const float* x, int64_t n, int32_t n_in, int32_t channels, int32_t n_out, const int32_t* src,
                            const float* w, float* out, trk_stream_t stream) {
    if (n < 0 || n_in < 1 || channels < 1 || n_out < 1 || !src || !w || (n > 0 && (!x || !out)))
        return fail(TRK_ERR_INVALID_ARG, "trk_interpolate_columns: bad argument");
    if (n == 0) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_interpolate_columns(x, n, n_in, channels, n_out, src, w, out, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_interpolate_columns_backward(const float* gout, int64_t n, int32_t n_in, int32_t channels, int32_t n_out, const int32_t* src,
                                     const float* w, float* gx, trk_stream_t stream) {
    if (n < 0 || n_in < 1 || channels < 1 || n_out < 1 || !src || !w || (n > 0 && (!gout || !gx)))
        return fail(TRK_ERR_INVALID_ARG, "trk_interpolate_columns_backward: bad argument");
    if (n == 0) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_interpolate_columns_bwd(gout, n, n_in, channels, n_out, src, w, gx, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_gp_prior_cost_grad(const void* q, const void* qd, int64_t batch, int32_t horizon, int32_t dof, int32_t io_dtype,
                           float dt, float sigma, float weight, float* cost, void* gq, void* gqd, int32_t grad_dtype,
                           float grad_scale, int32_t accumulate, trk_stream_t stream) {
    if (batch < 0 || horizon < 1 || dof < 1 || (io_dtype != TRK_F32 && io_dtype != TRK_F16) || !(dt > 0.0f) || !(sigma > 0.0f) ||
        batch > 0x7fffffff || (batch > 0 && (!q || !qd || !cost || !gq || !gqd)))
        return fail(TRK_ERR_INVALID_ARG, "trk_gp_prior_cost_grad: bad argument");
    if ((grad_dtype != TRK_F32 && grad_dtype != TRK_F16) || (io_dtype == TRK_F32 && grad_dtype == TRK_F16) || !(grad_scale > 0.0f) ||
        !std::isfinite(grad_scale))
        return fail(TRK_ERR_INVALID_ARG, "trk_gp_prior_cost_grad: grad_dtype must be TRK_F32 or (with fp16 trajectories) TRK_F16, grad_scale finite and > 0");
    if (batch == 0) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    if (trk_launch_gp_prior(io_dtype == TRK_F16, grad_dtype == TRK_F16, grad_scale, q, qd, batch, horizon, dof, dt, sigma, weight, cost, gq, gqd,
                            accumulate, (hipStream_t)stream))
        return fail(TRK_ERR_UNSUPPORTED, "trk_gp_prior_cost_grad: one trajectory (horizon x dof x 2 floats) must fit the 160 KiB LDS");
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_finite_difference(const float* x, int64_t batch, int32_t horizon, int32_t dim, float dt, int32_t method, float* out,
                          trk_stream_t stream) {
    if (batch < 0 || horizon < 1 || dim < 1 || method < 0 || method > 2 || !(dt != 0.0f) || (batch > 0 && (!x || !out)))
        return fail(TRK_ERR_INVALID_ARG, "trk_finite_difference: bad argument");
    if (batch == 0) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_finite_difference(x, batch, horizon, dim, dt, method, out, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_traj_diff_norm_sum(const float* x, int64_t batch, int32_t horizon, int32_t state_dim, int32_t c0, int32_t dim, float* out,
                           trk_stream_t stream) {
    if (batch < 0 || batch > 0x7fffffff || horizon < 1 || dim < 1 || c0 < 0 || c0 + dim > state_dim || (batch > 0 && (!x || !out)))
        return fail(TRK_ERR_INVALID_ARG, "trk_traj_diff_norm_sum: bad argument");
    if (batch == 0) return TRK_OK;
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_traj_diff_norm_sum(x, batch, horizon, state_dim, c0, dim, out, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_debug_set_stamp_buffer(void* device_u64) {
    g_stamps = static_cast<unsigned long long*>(device_u64);
    return TRK_OK;
}

int64_t trk_pack_sums_scratch_bytes(int32_t horizon, int32_t dof) {
    if (horizon < 1 || dof < 1) return TRK_ERR_INVALID_ARG;
    return (int64_t)(sizeof(float) * trk_pack_scratch_floats(horizon, dof));
}

int trk_pack_sums(const float* cost, const void* gq, int32_t grad_dtype, float grad_scale, const float* cost_block_sums,
                  const float* traj_cost, int64_t batch, int32_t horizon, int32_t dof, float* scratch, float* packed, trk_stream_t stream) {
    if (batch < 1 || batch > 0x7fffffff || horizon < 1 || dof < 1 || !cost || !gq || !cost_block_sums || !scratch || !packed ||
        (grad_dtype != TRK_F32 && grad_dtype != TRK_F16) || !(grad_scale > 0.0f) || !std::isfinite(grad_scale))
        return fail(TRK_ERR_INVALID_ARG, "trk_pack_sums: bad argument");
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_pack_sums(cost, gq, grad_dtype == TRK_F16, 1.0f / grad_scale, cost_block_sums, traj_cost, (int)batch, horizon, dof,
                         (batch * horizon + 63) / 64, scratch, packed, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_reduce_sum(const float* x, int64_t n, float* out, trk_stream_t stream) {
    if (n < 0 || !out || (n > 0 && !x)) return fail(TRK_ERR_INVALID_ARG, "trk_reduce_sum: bad argument");
    int rc = ensure_init();
    if (rc) return rc;
    trk_launch_reduce_sum(x, n, out, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_grid_precompute(const TrkCostModel* cm, const int32_t dims[3], const float lim_min[3], const float lim_max[3],
                        float* sdf, float* grad, trk_stream_t stream) {
    if (!cm || !dims || !lim_min || !lim_max || !sdf || !grad) return fail(TRK_ERR_INVALID_ARG, "trk_grid_precompute: null argument");
    for (int k = 0; k < 3; ++k) if (dims[k] < 1) return fail(TRK_ERR_INVALID_ARG, "trk_grid_precompute: bad dims");
    trk_launch_grid_precompute(cm->hdr, dims, lim_min, lim_max, sdf, grad, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

int trk_sdf_points(const TrkCostModel* cm, const float* points, int64_t n, float* sdf, float* grad, trk_stream_t stream) {
    if (!cm) return fail(TRK_ERR_INVALID_ARG, "trk_sdf_points: null cost model");
    if (n < 0 || (n > 0 && (!points || !sdf))) return fail(TRK_ERR_INVALID_ARG, "trk_sdf_points: bad argument");
    if (n == 0 || cm->hdr.n_objects == 0) return TRK_OK;
    trk_launch_sdf_points(cm->hdr, points, n, sdf, grad, (hipStream_t)stream);
    TRK_HIP(last_launch_error());
    return TRK_OK;
}

}  // extern "C"
