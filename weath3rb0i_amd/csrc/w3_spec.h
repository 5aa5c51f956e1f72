// w3_spec.h — host-side parsed form of w3_model_spec: the leaves in in-order (a BestOfTwo tree of any
// shape returns the leftmost leaf of maximal |p - 1/2|) followed by the APM chain applied at the root.
#pragma once
#include "../../include/w3hip.h"

struct ParsedSpec {
    int n_leaves = 0;
    w3_node leaf[W3_MAX_LEAVES];
    int n_apm = 0;
    w3_node apm[W3_MAX_APM];
    bool has_slot = false;
    uint32_t n_huff = 0;                  // HuffHistory table sets (caller's memory, valid during the call)
    const w3_huff_table *huff = nullptr;
    bool is_cm() const { return has_slot || n_apm > 0; }
};
