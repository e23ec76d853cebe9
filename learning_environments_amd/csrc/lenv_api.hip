// lenv_api.hip -- ABI housekeeping entry points of liblenv_hip.so (include/lenv_hip.h).
#include "lenv_device.cuh"

extern "C" int lenv_abi_version(void) { return LENV_ABI_VERSION; }

extern "C" int64_t lenv_struct_size(int32_t which)
{
    switch (which) {
    case 0: return sizeof(lenv_mlp_desc);
    case 1: return sizeof(lenv_ddqn_cfg);
    case 2: return sizeof(lenv_ql_cfg);
    case 3: return sizeof(lenv_td3_cfg);
    case 4: return sizeof(lenv_td3d_cfg);
    case 5: return sizeof(lenv_tapes);
    case 6: return sizeof(lenv_inner_out);
    case 7: return sizeof(lenv_ql_out);
    case 8: return sizeof(lenv_td3_tapes);
    case 9: return sizeof(lenv_td3_out);
    case 10: return sizeof(lenv_td3d_tapes);
    case 11: return sizeof(lenv_chain_hp);
    case 12: return sizeof(lenv_icm_io);
    default: return LENV_ERR_INVALID;
    }
}

extern "C" const char *lenv_error_string(int code)
{
    switch (code) {
    case LENV_OK: return "ok";
    case LENV_ERR_INVALID: return "invalid argument";
    case LENV_ERR_UNSUPPORTED: return "unsupported shape or option";
    case LENV_ERR_WORKSPACE: return "workspace too small";
    case LENV_ERR_LAUNCH: return "HIP launch failed";
    case LENV_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown error";
    }
}

// same key schedule as the oracle's orc_chain_key
// one uniform in [0, 1) of a chain's counter RNG (host): the *_vary hyper-parameter draws (STREAM_VARY_HP) are made on the host
extern "C" double lenv_rng_unit(uint64_t key, uint32_t stream, uint64_t index)
{
    return (double)(lenv::rng_u64(key, stream, index) >> 11) * (1.0 / 9007199254740992.0);
}

extern "C" uint64_t lenv_chain_key(uint64_t seed, uint64_t generation, uint64_t worker, uint64_t kind)
{
    uint64_t k = lenv::mix64(seed + 0x9e3779b97f4a7c15ULL);
    k = lenv::mix64(k ^ (generation + 0x9e3779b97f4a7c15ULL * 2));
    k = lenv::mix64(k ^ (worker + 0x9e3779b97f4a7c15ULL * 3));
    k = lenv::mix64(k ^ (kind + 0x9e3779b97f4a7c15ULL * 4));     // own round: (worker, kind) never collides with (worker + 1, kind - 4)
    return k;
}
