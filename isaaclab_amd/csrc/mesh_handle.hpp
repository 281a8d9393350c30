// Host-side handle behind mdp_mesh_t (opaque at the C ABI).
#pragma once
#include <cstdint>
#include <vector>
#include "raycast_device.hpp"

struct mdp_mesh {
    mdp_mesh_view_t view{};      // device pointers + grid / BVH geometry (passed to kernels by value)
    int device = 0;
    int64_t num_tris = 0;
    int64_t num_cell_refs = 0;
    int64_t num_patch_verts = 0;       // vertices held by lattice patches (0: none found)
    int64_t device_bytes = 0;
    std::vector<int32_t> h_cell_tris;  // build-time scratch
};
