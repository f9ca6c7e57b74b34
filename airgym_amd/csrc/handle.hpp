// handle.hpp - what an ag_handle points to (private to the library: airgym_hip.hip and, in the experiments build,
// experiments.hip).  The public view is include/airgym_hip.h.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "../../include/airgym_hip.h"
#include "kernel_args.hpp"

struct AgLayout {
    size_t S[4], C[4], PA, PA4, obs, rew, reset, timeout, mask, reset_ids, reset_count, tick, terms[11], cmd, total;
    size_t OB, GOAL, PRP, image, collisions, table;   // planning only
};


struct ag_env {
    ag_config cfg;
    int num_obs, num_actions, n_pad;
    char* arena;
    bool owns_arena;
    AgLayout L;
    ag::KArgs k;       // pointers + StepParams template for launches
    ag::PlanArgs pa;   // planning extras
    bool table_set;    // planning: obstacle variant table uploaded
    uint64_t counter;  // planning: pre_physics_step counter driving the camera schedule (planning.py:153-156)
    int force_render;  // planning: render on the next step regardless of the schedule
    int last_rendered; // planning / avoid: 1 if the most recent step produced a new depth image
    uint64_t tick;     // host mirror of the device tick (exact unless a captured graph is being replayed)
    int parity;        // which of the two device tick slots the next launch reads
};

