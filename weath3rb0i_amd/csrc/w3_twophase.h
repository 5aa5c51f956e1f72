// w3_twophase.h — placeholder until the two-phase kernels land (next commit).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "w3_spec.h"

struct TwoPhaseWs { void release() {} };
static inline bool twophase_supported(const ParsedSpec &, size_t) { return false; }
static inline int twophase_encode(TwoPhaseWs &, hipStream_t, const ParsedSpec &, const uint8_t *, size_t, size_t, uint32_t, uint8_t *,
                                  uint32_t, uint32_t *, uint32_t *, hipEvent_t *, w3_timing *, std::string &) { return W3_E_UNSUPPORTED; }
static inline int twophase_predict(TwoPhaseWs &, hipStream_t, const ParsedSpec &, const uint8_t *, size_t, size_t, uint32_t,
                                   const uint16_t **, hipEvent_t *, w3_timing *, std::string &) { return W3_E_UNSUPPORTED; }
