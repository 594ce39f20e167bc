// api_internal.hpp -- what the three host files behind the C ABI share: api.hip (context, options, meshes, box lists),
// build.hip (BroadPhase: cell grid, entry lists, sort, records, speculative build, sweep + overflow / cursor, sharding
// windows) and drivers.hip (narrow_phase, ccd(), ccd() with collisions, ipc_ccd_strategy).  Host code only.
#pragma once
#include "internal.hpp"
#include "grid.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------
// error trampolines
extern thread_local std::string g_create_error; // (api.hip)

template <class Fn> int guarded(sccd_ctx* c, Fn&& fn)
{
    try {
        if (c) SCCD_HIP(hipSetDevice(c->device));
        fn();
        return SCCD_OK;
    } catch (const SccdError& e) {
        (void)hipGetLastError(); // (sticky: a later launch check must not trip over this call's failure)
        if (c) c->err = e.msg;
        else g_create_error = e.msg;
        return e.code;
    } catch (const std::bad_alloc&) {
        if (c) c->err = "host allocation failed";
        return SCCD_E_NOMEM;
    } catch (const std::exception& e) {
        if (c) c->err = e.what();
        return SCCD_E_INVALID;
    }
}

// pipeline objects cached in the context so that repeated ccd() calls allocate nothing
struct Pipeline {
    sccd_boxes vb, eb, fb; // boxes in element order (raw)
    sccd_broad_phase bp;
    sccd_broad_phase bp_ee; // edge-edge lists of ccd(): belongs to the helper context c->side
};
Pipeline* pipeline_of(sccd_ctx* c); // (api.hip)

// api.hip
void merge_side_profile(sccd_ctx* c);
void copy_in(sccd_ctx* c, void* dst, const void* src, size_t bytes, int src_on_device);
sccd_mesh* scratch_mesh_from_host(sccd_ctx* c, const double* V0, const double* V1, int nV, const int32_t* E, int nE, const int32_t* F,
                                  int nF, bool defer_verdict = false);
void mesh_deferred_verdict(sccd_ctx* c);
void boxes_from_mesh(sccd_ctx* c, const sccd_mesh* m, double r, Pipeline* pl, bool want_v, bool want_e, bool want_f, bool lazy_ef = false,
                     hipEvent_t after_vertices = nullptr);
void edge_boxes_on(sccd_ctx* sc, const sccd_mesh* m, Pipeline* pl);
void ensure_stats(sccd_ctx* c, const sccd_boxes* b);
// build.hip
void bp_build(sccd_broad_phase* bp, const sccd_boxes* A, const sccd_boxes* B);
void bp_detect_partial(sccd_broad_phase* bp, int phase = 0);
