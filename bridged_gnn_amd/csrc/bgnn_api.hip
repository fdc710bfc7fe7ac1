#include "bgnn_common.h"

extern "C" int bgnn_version(void) { return BGNN_VERSION; }

#ifndef BGNN_SRC_HASH
#define BGNN_SRC_HASH "unknown"
#endif
extern "C" const char* bgnn_source_hash(void) { return BGNN_SRC_HASH; }

extern "C" const char* bgnn_error_string(int code) {
  switch (code) {
    case 0: return "success";
    case BGNN_E_NULL: return "BGNN_E_NULL: required pointer is NULL";
    case BGNN_E_SHAPE: return "BGNN_E_SHAPE: unsupported or inconsistent shape";
    case BGNN_E_WORKSPACE: return "BGNN_E_WORKSPACE: workspace too small";
    case BGNN_E_ALIGN: return "BGNN_E_ALIGN: pointer / leading dimension not 16-byte aligned";
    case BGNN_E_RANGE: return "BGNN_E_RANGE: k or index range not supported";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown bgnn error";
  }
}
