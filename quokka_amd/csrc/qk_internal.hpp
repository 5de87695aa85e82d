// qk_internal.hpp — host-side internals shared by the translation units of libquokka_amd.so.
#ifndef QK_INTERNAL_HPP_
#define QK_INTERNAL_HPP_

#include <hip/hip_runtime.h>

#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "quokka_amd.h"

struct qk_prof_pending {
	int slot;
	hipEvent_t start, stop;
};
struct qk_prof_slot {
	std::string name;
	long count = 0;
	double total_ms = 0.0;
};

struct qk_ctx {
	int device = 0;
	std::string last_error;
	std::vector<void *> owned; // device allocations released in qk_ctx_destroy
	std::mutex mtx;
	// optional per-kernel HIP-event timing (qk_profile_*): events are recorded on the launch stream
	bool profiling = false;
	std::string prof_only; // qk_profile_only: time this kernel alone (empty: all)
	std::vector<qk_prof_slot> prof_slots;
	std::vector<qk_prof_pending> prof_pending;
	std::vector<hipEvent_t> prof_free_events;
	int *counter_slots = nullptr; // qk_rad_ops.hip: spread iteration / failure counters (owned)
};

struct qk_level {
	qk_ctx *ctx = nullptr;
	int ndim = 3;
	int nboxes = 0;
	std::vector<qk_box> boxes; // host copy
	qk_box *d_boxes = nullptr; // device copy
	int maxlen[3] = {0, 0, 0}; // max valid-box length per dim
	// cache of the fused path (ghost-4 scratch geometry of every box), built on first use
	void *d_sgeom = nullptr;
	int64_t sgeom_total_cells = 0;
	// device buffer of qk_amr_tile_flags (one int per blocking-factor tile of the domain), kept between regrids
	int *d_tile_flags = nullptr;
	size_t tile_flags_bytes = 0;
	// qk_cooling.hip: the next-cell counter of the persistent kernel — one per level (a level is used from one stream at a time,
	// include/quokka_amd.h: two levels cooling on two streams each clear and count their own word)
	unsigned long long *d_cooling_queue = nullptr;
};

namespace qk
{

inline auto setError(qk_ctx *ctx, int code, const char *what, const char *detail = "") -> int
{
	if (ctx != nullptr) {
		std::lock_guard<std::mutex> lock(ctx->mtx);
		ctx->last_error = std::string(what) + (detail[0] != 0 ? ": " : "") + detail;
	}
	return code;
}

#define QK_HIP_CHECK(ctx, expr)                                                                                                                      \
	do {                                                                                                                                         \
		hipError_t qk_e_ = (expr);                                                                                                           \
		if (qk_e_ != hipSuccess) {                                                                                                           \
			return qk::setError((ctx), QK_ERR_HIP, #expr, hipGetErrorString(qk_e_));                                                     \
		}                                                                                                                                    \
	} while (0)

#define QK_REQUIRE(ctx, cond, msg)                                                                                                                   \
	do {                                                                                                                                         \
		if (!(cond)) {                                                                                                                       \
			return qk::setError((ctx), QK_ERR_INVALID, msg);                                                                             \
		}                                                                                                                                    \
	} while (0)

inline auto checkTraits(qk_ctx *ctx, const qk_hydro_traits *t) -> int
{
	if (t == nullptr) {
		return setError(ctx, QK_ERR_INVALID, "traits is NULL");
	}
	if (t->nscalars < 0 || t->nscalars > QK_MAX_SCALARS || t->nmscalars < 0 || t->nmscalars > t->nscalars) {
		return setError(ctx, QK_ERR_UNSUPPORTED, "0..QK_MAX_SCALARS passive scalars, of which the first nmscalars (0..nscalars) are mass scalars");
	}
	if (t->eos_temperature_model != QK_HOOK_COMPILED &&
	    (t->eos_temperature_model < 0 || t->eos_temperature_model > 1 || (t->eos_temperature_model == 1 && !(t->eos_alpha > 0.0)))) {
		return setError(ctx, QK_ERR_UNSUPPORTED, "eos_temperature_model must be 0 (gamma law), 1 (E = alpha / 4 T^4, alpha > 0) or QK_HOOK_COMPILED");
	}
	if (t->ndim < 1 || t->ndim > 3) {
		return setError(ctx, QK_ERR_UNSUPPORTED, "ndim must be 1, 2 or 3");
	}
	return QK_OK;
}

// an entry point that evaluates the temperature hooks of quokka::EOS cannot serve a problem whose hooks are compiled device code
inline auto needsLibraryEos(qk_ctx *ctx, const qk_hydro_traits *t, const char *who) -> int
{
	if (t->eos_temperature_model == QK_HOOK_COMPILED) {
		return setError(ctx, QK_ERR_UNSUPPORTED, who,
				"the quokka::EOS temperature hooks of this problem are compiled device code: instantiate the kernel in the problem's translation unit "
				"(quokka_amd/host/qk_problem_kernels.hpp)");
	}
	return QK_OK;
}

// launch geometry for "one thread per cell of box b grown by ng (+1 face in direction facedir)"
struct CellLaunch {
	dim3 grid;
	dim3 block;
};

inline auto cellLaunch(const qk_level *lev, int ng, int facedir) -> CellLaunch
{
	int64_t n = 1;
	for (int d = 0; d < 3; ++d) {
		int len = lev->maxlen[d];
		if (d < lev->ndim) {
			len += 2 * ng;
		}
		if (d == facedir) {
			len += 1;
		}
		n *= len;
	}
	CellLaunch L;
	L.block = dim3(256, 1, 1);
	L.grid = dim3(static_cast<unsigned>((n + 255) / 256), static_cast<unsigned>(lev->nboxes), 1);
	return L;
}

// RAII scope: records a start/stop HIP event pair around the launches issued inside it (no-op unless profiling)
struct ProfScope {
	qk_ctx *ctx;
	hipStream_t s;
	int idx = -1;
	ProfScope(qk_ctx *c, hipStream_t stream, const char *name) : ctx(c), s(stream)
	{
		if (ctx == nullptr || !ctx->profiling || (!ctx->prof_only.empty() && ctx->prof_only != name)) {
			return;
		}
		int slot = -1;
		for (size_t i = 0; i < ctx->prof_slots.size(); ++i) {
			if (ctx->prof_slots[i].name == name) {
				slot = static_cast<int>(i);
			}
		}
		if (slot < 0) {
			ctx->prof_slots.push_back({name, 0, 0.0});
			slot = static_cast<int>(ctx->prof_slots.size()) - 1;
		}
		auto getEvent = [&]() {
			hipEvent_t e = nullptr;
			if (!ctx->prof_free_events.empty()) {
				e = ctx->prof_free_events.back();
				ctx->prof_free_events.pop_back();
			} else {
				(void)hipEventCreate(&e);
			}
			return e;
		};
		qk_prof_pending p{slot, getEvent(), getEvent()};
		(void)hipEventRecord(p.start, s);
		ctx->prof_pending.push_back(p);
		idx = static_cast<int>(ctx->prof_pending.size()) - 1;
	}
	~ProfScope()
	{
		if (idx >= 0) {
			(void)hipEventRecord(ctx->prof_pending[idx].stop, s);
		}
	}
};

} // namespace qk

#endif // QK_INTERNAL_HPP_
