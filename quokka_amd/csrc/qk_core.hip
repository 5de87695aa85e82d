// qk_core.hip — context, level (BoxArray) and descriptor-table plumbing of the C-ABI (include/quokka_amd.h).
#include <algorithm>
#include <cstring>

#include "qk_internal.hpp"

extern "C" {

const char *qk_version(void) { return "quokka_amd 0.1 (gfx950, fp64, ffp-contract=off)"; }

int qk_ctx_create(qk_ctx **ctx, int device)
{
	if (ctx == nullptr) {
		return QK_ERR_INVALID;
	}
	if (device == QK_DEVICE_HOST_PLANNING) {
		// planning-only context: box/ghost-plan logic (pure host code) without a GPU; every kernel entry point
		// fails with QK_ERR_HIP on such a context.  Used by the multi-rank CPU (gloo) tests of the exchange protocol.
		auto *c = new qk_ctx;
		c->device = -1;
		*ctx = c;
		return QK_OK;
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
		// fail loudly: there is no CPU fallback in the product path
		std::fprintf(stderr, "quokka_amd: no HIP device visible; the hot path has no CPU fallback\n");
		return QK_ERR_HIP;
	}
	if (device < 0 || device >= ndev) {
		return QK_ERR_INVALID;
	}
	if (hipSetDevice(device) != hipSuccess) {
		return QK_ERR_HIP;
	}
	auto *c = new qk_ctx;
	c->device = device;
	*ctx = c;
	return QK_OK;
}

int qk_ctx_destroy(qk_ctx *ctx)
{
	if (ctx == nullptr) {
		return QK_ERR_INVALID;
	}
	for (void *p : ctx->owned) {
		(void)hipFree(p);
	}
	delete ctx;
	return QK_OK;
}

const char *qk_last_error(qk_ctx *ctx)
{
	if (ctx == nullptr) {
		return "null context";
	}
	return ctx->last_error.c_str();
}

int qk_level_create(qk_ctx *ctx, qk_level **lev, int ndim, int nboxes, const qk_box *valid_boxes)
{
	if (ctx == nullptr || lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(ctx, nboxes >= 0 && (valid_boxes != nullptr || nboxes == 0), "qk_level_create: bad box list");
	// nboxes == 0 is a rank that owns no box of this level (AMR): every operator on such a level is a no-op
	QK_REQUIRE(ctx, ndim >= 1 && ndim <= 3, "qk_level_create: ndim must be 1, 2 or 3");
	auto *L = new qk_level;
	L->ctx = ctx;
	L->ndim = ndim;
	L->nboxes = nboxes;
	L->boxes.assign(valid_boxes, valid_boxes + nboxes);
	for (int b = 0; b < nboxes; ++b) {
		for (int d = 0; d < 3; ++d) {
			L->maxlen[d] = std::max(L->maxlen[d], valid_boxes[b].hi[d] - valid_boxes[b].lo[d] + 1);
		}
	}
	if (ctx->device < 0) { // planning-only context: no device copy
		*lev = L;
		return QK_OK;
	}
	hipError_t e = hipMalloc(reinterpret_cast<void **>(&L->d_boxes), sizeof(qk_box) * (nboxes > 0 ? nboxes : 1));
	if (e == hipSuccess) {
		e = (nboxes > 0) ? hipMemcpy(L->d_boxes, valid_boxes, sizeof(qk_box) * nboxes, hipMemcpyHostToDevice) : hipSuccess;
	}
	if (e != hipSuccess) {
		delete L;
		return qk::setError(ctx, QK_ERR_HIP, "qk_level_create", hipGetErrorString(e));
	}
	*lev = L;
	return QK_OK;
}

int qk_level_destroy(qk_level *lev)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	(void)hipFree(lev->d_boxes);
	(void)hipFree(lev->d_sgeom);
	(void)hipFree(lev->d_tile_flags);
	(void)hipFree(lev->d_cooling_queue);
	delete lev;
	return QK_OK;
}

int qk_clear_bytes(qk_ctx *ctx, qk_stream s, void *device_ptr, int64_t nbytes)
{
	if (ctx == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(ctx, device_ptr != nullptr && nbytes >= 0, "qk_clear_bytes: NULL pointer or negative size");
	QK_HIP_CHECK(ctx, hipMemsetAsync(device_ptr, 0, static_cast<size_t>(nbytes), static_cast<hipStream_t>(s)));
	return QK_OK;
}

int qk_profile_enable(qk_ctx *ctx, int on)
{
	if (ctx == nullptr) {
		return QK_ERR_INVALID;
	}
	ctx->profiling = (on != 0);
	return QK_OK;
}

int qk_profile_only(qk_ctx *ctx, const char *kernel_name)
{
	if (ctx == nullptr) {
		return QK_ERR_INVALID;
	}
	ctx->prof_only = (kernel_name != nullptr) ? kernel_name : "";
	return QK_OK;
}

// folds finished event pairs into the per-kernel totals (synchronises on the recorded events)
static void profCollect(qk_ctx *ctx)
{
	for (auto &p : ctx->prof_pending) {
		float ms = 0.f;
		if (hipEventSynchronize(p.stop) == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
			ctx->prof_slots[p.slot].count += 1;
			ctx->prof_slots[p.slot].total_ms += ms;
		}
		ctx->prof_free_events.push_back(p.start);
		ctx->prof_free_events.push_back(p.stop);
	}
	ctx->prof_pending.clear();
}

int qk_profile_reset(qk_ctx *ctx)
{
	if (ctx == nullptr) {
		return QK_ERR_INVALID;
	}
	profCollect(ctx);
	ctx->prof_slots.clear();
	return QK_OK;
}

int qk_profile_num_kernels(qk_ctx *ctx)
{
	if (ctx == nullptr) {
		return QK_ERR_INVALID;
	}
	profCollect(ctx);
	return static_cast<int>(ctx->prof_slots.size());
}

int qk_profile_get(qk_ctx *ctx, int k, const char **name, long *count, double *total_ms)
{
	if (ctx == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(ctx, k >= 0 && k < static_cast<int>(ctx->prof_slots.size()) && name && count && total_ms, "qk_profile_get: bad index");
	*name = ctx->prof_slots[k].name.c_str();
	*count = ctx->prof_slots[k].count;
	*total_ms = ctx->prof_slots[k].total_ms;
	return QK_OK;
}

static int uploadTable(qk_ctx *ctx, int n, const void *host, size_t elem, void **dev)
{
	if (ctx == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(ctx, host != nullptr && dev != nullptr && n > 0, "qk_upload_*_table: bad arguments");
	void *p = nullptr;
	QK_HIP_CHECK(ctx, hipMalloc(&p, elem * n));
	QK_HIP_CHECK(ctx, hipMemcpy(p, host, elem * n, hipMemcpyHostToDevice));
	{
		std::lock_guard<std::mutex> lock(ctx->mtx);
		ctx->owned.push_back(p);
	}
	*dev = p;
	return QK_OK;
}

int qk_upload_array4_table(qk_ctx *ctx, int n, const qk_array4 *host_table, qk_array4 **device_table)
{
	return uploadTable(ctx, n, host_table, sizeof(qk_array4), reinterpret_cast<void **>(device_table));
}

int qk_upload_iarray4_table(qk_ctx *ctx, int n, const qk_iarray4 *host_table, qk_iarray4 **device_table)
{
	return uploadTable(ctx, n, host_table, sizeof(qk_iarray4), reinterpret_cast<void **>(device_table));
}

} // extern "C"
