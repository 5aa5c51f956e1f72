// w3_spec.h — host-side parsed form of w3_model_spec: the leaves in in-order.
#pragma once
#include "../../include/w3hip.h"

struct ParsedSpec {
    int n_leaves = 0;
    w3_node leaf[W3_MAX_LEAVES];
};
