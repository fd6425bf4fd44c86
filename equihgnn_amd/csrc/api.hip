// Library identity + error strings.
#include "common.h"

extern "C" int eqh_version(void) { return 1; }

extern "C" const char* eqh_error_string(int code) {
    switch (code) {
        case EQH_OK: return "ok";
        case EQH_ERR_ARG: return "invalid argument (null pointer, negative size or unsupported shape)";
        case EQH_ERR_ALIGN: return "row length not a multiple of 4 floats or pointer not 16-byte aligned";
        case EQH_ERR_RANGE: return "size exceeds int32 indexing";
        case EQH_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}
