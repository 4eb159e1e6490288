/* oc_hostio.h -- C ABI of liboc_hostio.so: everything the SB3-shaped numpy boundary returns for
 * one step, packed by ONE launch into one device buffer that crosses PCIe as one copy.
 *
 * What it replaces: the per-step obs dict -> numpy conversion a stable-baselines3 VecEnv consumer
 * sees (`VecEnv.step_wait() -> (obs dict of [n, k] arrays in the declared space dtypes, rewards,
 * dones, infos)`; reference: `DummyVecEnv` around `OvercookedMultiEnv`, trainer.py:87-121, spaces
 * gym_comm/envs/overcooked_env.py:41-85).  The stepper leaves a viewer's observation as [F][n]
 * rows; the numpy API wants [n, k] arrays of int64 / float32 / int8 per key.  `oc_pack_host`
 * gathers the rows of each dtype group, transposes, converts, and appends the float32 timestep,
 * reward, episode return, the int32 done flags and episode lengths.
 *
 * Output buffer (bytes, every block aligned to its element size because the widest come first):
 *   int64   [n][w64]     rows whose plan entry names block 0, at their column
 *   float64 [n]          ep_return   (present iff ep_return != NULL; Monitor rounds it to 6 decimals)
 *   float32 [n][w32]     block 1
 *   float32 [n]          timestep
 *   float32 [n]          reward      (present iff reward != NULL)
 *   int32   [n]          done        (present iff done != NULL)
 *   int32   [n]          ep_length   (present iff ep_length != NULL)
 *   int8    [n][w8]      block 2
 * `oc_pack_host_bytes` returns the total for the same argument presence flags.
 * plan: DEVICE int32 [F]: for observation row r, (block << 16) | column.
 * All pointers are device pointers of caller-owned tensors; nothing is allocated or synchronised. */
#ifndef OC_HOSTIO_H
#define OC_HOSTIO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#ifndef OC_API
#define OC_API __attribute__((visibility("default")))
#endif

#define OC_HOSTIO_ABI_VERSION 1

OC_API int oc_hostio_abi_version(void);
OC_API const char *oc_hostio_last_error(void);
OC_API int64_t oc_pack_host_bytes(int32_t w64, int32_t w32, int32_t w8, int32_t has_reward, int32_t has_ep_return,
                                  int32_t has_done, int32_t has_ep_length, int64_t n);
/* obs_type: element type of the rows, 0 int32, 1 int8, 2 float32 (oc_obs_cfg.obs_int8) */
OC_API int oc_pack_host(const void *obs_rows, int32_t obs_type, int32_t F, const int32_t *plan, int32_t w64,
                        int32_t w32, int32_t w8, const double *timestep, const double *reward,
                        const double *ep_return, const int32_t *done, const int32_t *ep_length, void *out,
                        int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif
