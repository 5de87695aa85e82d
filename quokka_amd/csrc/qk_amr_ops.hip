// qk_amr_ops.hip — the data-parallel pieces of the AMR level machinery, one kernel each:
//   qk_tag_relative_gradient   the gradient-threshold ErrorEst of the reference's problems
//                              (src/problems/HydroBlast3D/test_hydro3d_blast.cpp:118-151, RadhydroShell/test_radhydro_shell.cpp:337-371)
//   qk_average_down            AMRSimulation::AverageDownTo -> amrex::average_down (src/simulation.hpp:1949-1964), cell-centred,
//                              conservative mean of the ratio^3 fine cells under a coarse cell
//   qk_PreInterpState / qk_PostInterpState   QuokkaSimulation::PreInterpState / PostInterpState (src/QuokkaSimulation.hpp:804-841)
// Grid generation, the FillPatch interpolation itself, flux registers and the subcycling driver are not built yet.
#include <algorithm>
#include <vector>

#include "qk_device.hpp"
#include "qk_internal.hpp"

using namespace qk;

struct AvgItem {
	int crse_box, fine_box;
	int lo[3], hi[3]; // region in COARSE index space
};

struct qk_avgdown_plan {
	qk_level *crse = nullptr;
	qk_level *fine = nullptr;
	int ratio[3] = {2, 2, 2};
	std::vector<AvgItem> items;
	AvgItem *d_items = nullptr;
	int64_t max_cells = 0;
};

namespace
{

using CA4 = A4<char, qk_carray4>;

template <class F> __global__ void __launch_bounds__(256) k_valid_cells(const qk_box *boxes, F f)
{
	const int b = blockIdx.y;
	const qk_box bx = boxes[b];
	const int l0 = bx.hi[0] - bx.lo[0] + 1, l1 = bx.hi[1] - bx.lo[1] + 1, l2 = bx.hi[2] - bx.lo[2] + 1;
	const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (t >= static_cast<int64_t>(l0) * l1 * l2) {
		return;
	}
	const int k = static_cast<int>(t / (static_cast<int64_t>(l0) * l1));
	const int r = static_cast<int>(t - static_cast<int64_t>(k) * l0 * l1);
	const int j = r / l0;
	f(b, bx.lo[0] + (r - j * l0), bx.lo[1] + j, bx.lo[2] + k);
}

__global__ void __launch_bounds__(256) k_average_down(const AvgItem *items, const qk_array4 *fine_t, qk_array4 *crse_t, int scomp, int ncomp, int r0, int r1,
						      int r2)
{
	const AvgItem it = items[blockIdx.y];
	const int n0 = it.hi[0] - it.lo[0] + 1, n1 = it.hi[1] - it.lo[1] + 1, n2 = it.hi[2] - it.lo[2] + 1;
	const int64_t ncell = static_cast<int64_t>(n0) * n1 * n2;
	RA4 F(fine_t[it.fine_box]);
	WA4 Cc(crse_t[it.crse_box]);
	// amrex_avgdown: volfrac = 1 / (rx ry rz);  c = sum over kref, jref, iref (iref fastest);  crse = volfrac * c
	const double volfrac = 1.0 / static_cast<double>(r0 * r1 * r2);
	for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < ncell * ncomp; t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
		const int n = static_cast<int>(t / ncell);
		const int64_t c = t - n * ncell;
		const int k = static_cast<int>(c / (static_cast<int64_t>(n0) * n1));
		const int r = static_cast<int>(c - static_cast<int64_t>(k) * n0 * n1);
		const int j = r / n0;
		const int i = it.lo[0] + (r - j * n0), jj = it.lo[1] + j, kk = it.lo[2] + k;
		double sum = 0.0;
		for (int kr = 0; kr < r2; ++kr) {
			for (int jr = 0; jr < r1; ++jr) {
				for (int ir = 0; ir < r0; ++ir) {
					sum += F(i * r0 + ir, jj * r1 + jr, kk * r2 + kr, scomp + n);
				}
			}
		}
		Cc(i, jj, kk, scomp + n) = volfrac * sum;
	}
}

} // namespace

extern "C" {

int qk_tag_relative_gradient(qk_level *lev, qk_stream s, const qk_hydro_traits *t, const qk_array4 *state_t, qk_carray4 *tags_t, int field, double eta_threshold,
			     double q_min, int min_inclusive)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	if (int rc = checkTraits(lev->ctx, t); rc != QK_OK) {
		return rc;
	}
	QK_REQUIRE(lev->ctx, state_t && tags_t, "tag_relative_gradient: NULL array");
	QK_REQUIRE(lev->ctx, field == QK_TAGFIELD_PRESSURE || (field >= 0 && field < 64), "tag_relative_gradient: unknown field");
	if (lev->nboxes == 0) {
		return QK_OK;
	}
	const Eos eos(*t);
	const int ndim = lev->ndim;
	int64_t maxcells = 1;
	for (int d = 0; d < 3; ++d) {
		maxcells *= lev->maxlen[d];
	}
	const dim3 grid(static_cast<unsigned>((maxcells + 255) / 256), static_cast<unsigned>(lev->nboxes), 1);
	auto f = [=] __device__(int b, int i, int j, int k) {
		RA4 U(state_t[b]);
		CA4 tag(tags_t[b]);
		auto q = [&](int ii, int jj, int kk) -> double {
			if (field == QK_TAGFIELD_PRESSURE) { // HydroSystem::ComputePressure(state, i, j, k)
				return consPressure(eos, U(ii, jj, kk, RHO), U(ii, jj, kk, MX), U(ii, jj, kk, MY), U(ii, jj, kk, MZ), U(ii, jj, kk, ENE));
			}
			return U(ii, jj, kk, field);
		};
		const double c = q(i, j, k);
		const bool above = (min_inclusive != 0) ? (c >= q_min) : (c > q_min);
		if (!above) { // (the tag needs both conditions: a cell below the threshold never reads its neighbours — ambient gas costs one pressure, not seven)
			return;
		}
		// del_d = max(|q+ - q|, |q - q-|);  indicator = max(del_x, del_y, del_z) / q   (std::max: first argument on ties)
		double del = smax(fabs(q(i + 1, j, k) - c), fabs(c - q(i - 1, j, k)));
		if (ndim >= 2) {
			del = smax(del, smax(fabs(q(i, j + 1, k) - c), fabs(c - q(i, j - 1, k))));
		}
		if (ndim == 3) {
			del = smax(del, smax(fabs(q(i, j, k + 1) - c), fabs(c - q(i, j, k - 1))));
		}
		const double gradient_indicator = del / c;
		if (gradient_indicator > eta_threshold) {
			tag(i, j, k) = static_cast<char>(QK_TAG_SET);
		}
	};
	hipLaunchKernelGGL(k_valid_cells<decltype(f)>, grid, dim3(256), 0, static_cast<hipStream_t>(s), lev->d_boxes, f);
	QK_HIP_CHECK(lev->ctx, hipGetLastError());
	return QK_OK;
}

// ErrorEst of HydroShocktube (src/problems/HydroShocktube/test_hydro_shocktube.cpp:146-170): centred difference along one direction,
//   del = (q(i+1) - q(i-1)) / (2 dx);  indicator = sqrt(del * del) / q;  SET where indicator > eta and q >= q_min (> if !min_inclusive)
int qk_tag_centered_gradient(qk_level *lev, qk_stream s, const qk_array4 *state_t, qk_carray4 *tags_t, int comp, int dir, double dx, double eta_threshold,
			     double q_min, int min_inclusive)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(lev->ctx, state_t && tags_t && comp >= 0 && dir >= 0 && dir < lev->ndim && dx > 0.0, "tag_centered_gradient: bad argument");
	if (lev->nboxes == 0) {
		return QK_OK;
	}
	int64_t maxcells = 1;
	for (int d = 0; d < 3; ++d) {
		maxcells *= lev->maxlen[d];
	}
	const dim3 grid(static_cast<unsigned>((maxcells + 255) / 256), static_cast<unsigned>(lev->nboxes), 1);
	auto f = [=] __device__(int b, int i, int j, int k) {
		RA4 U(state_t[b]);
		CA4 tag(tags_t[b]);
		const int ex = (dir == 0), ey = (dir == 1), ez = (dir == 2);
		const double q = U(i, j, k, comp);
		const double del = (U(i + ex, j + ey, k + ez, comp) - U(i - ex, j - ey, k - ez, comp)) / (2.0 * dx);
		const double gradient_indicator = sqrt(del * del) / q;
		const bool above = (min_inclusive != 0) ? (q >= q_min) : (q > q_min);
		if (gradient_indicator > eta_threshold && above) {
			tag(i, j, k) = static_cast<char>(QK_TAG_SET);
		}
	};
	hipLaunchKernelGGL(k_valid_cells<decltype(f)>, grid, dim3(256), 0, static_cast<hipStream_t>(s), lev->d_boxes, f);
	QK_HIP_CHECK(lev->ctx, hipGetLastError());
	return QK_OK;
}

// PreInterpState / PostInterpState: total energy <-> specific internal energy around the coarse-to-fine interpolation
static int interpState(qk_level *lev, qk_stream s, qk_array4 *mf_t, bool pre)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(lev->ctx, mf_t, "Pre/PostInterpState: NULL array");
	if (lev->nboxes == 0) {
		return QK_OK;
	}
	int64_t maxcells = 1;
	for (int d = 0; d < 3; ++d) {
		maxcells *= lev->maxlen[d];
	}
	const dim3 grid(static_cast<unsigned>((maxcells + 255) / 256), static_cast<unsigned>(lev->nboxes), 1);
	auto f = [=] __device__(int b, int i, int j, int k) {
		WA4 cons(mf_t[b]);
		const double rho = cons(i, j, k, RHO);
		const double px = cons(i, j, k, MX);
		const double py = cons(i, j, k, MY);
		const double pz = cons(i, j, k, MZ);
		const double kinetic_energy = (px * px + py * py + pz * pz) / (2.0 * rho);
		if (pre) {
			const double Etot = cons(i, j, k, ENE);
			cons(i, j, k, ENE) = (Etot - kinetic_energy) / rho; // specific internal energy
		} else {
			const double e = cons(i, j, k, ENE);
			const double Eint = rho * e;
			cons(i, j, k, ENE) = Eint + kinetic_energy;
		}
	};
	hipLaunchKernelGGL(k_valid_cells<decltype(f)>, grid, dim3(256), 0, static_cast<hipStream_t>(s), lev->d_boxes, f);
	QK_HIP_CHECK(lev->ctx, hipGetLastError());
	return QK_OK;
}

int qk_PreInterpState(qk_level *lev, qk_stream s, qk_array4 *mf) { return interpState(lev, s, mf, true); }
int qk_PostInterpState(qk_level *lev, qk_stream s, qk_array4 *mf) { return interpState(lev, s, mf, false); }

int qk_avgdown_plan_create(qk_level *crse, qk_level *fine, const int ratio[3], qk_avgdown_plan **plan)
{
	if (crse == nullptr || fine == nullptr || plan == nullptr || ratio == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = crse->ctx;
	QK_REQUIRE(ctx, crse->ctx == fine->ctx && crse->ndim == fine->ndim, "avgdown_plan_create: levels of different contexts / dimensions");
	auto *P = new qk_avgdown_plan;
	P->crse = crse;
	P->fine = fine;
	for (int d = 0; d < 3; ++d) {
		P->ratio[d] = (d < crse->ndim) ? ratio[d] : 1;
		if (P->ratio[d] < 1) {
			delete P;
			return setError(ctx, QK_ERR_INVALID, "avgdown_plan_create: refinement ratio < 1");
		}
	}
	auto fdiv = [](int a, int r) { return (a >= 0) ? a / r : -((-a + r - 1) / r); }; // amrex::coarsen (floor division)
	for (int f = 0; f < fine->nboxes; ++f) {
		qk_box cf{};
		for (int d = 0; d < 3; ++d) {
			cf.lo[d] = fdiv(fine->boxes[f].lo[d], P->ratio[d]);
			cf.hi[d] = fdiv(fine->boxes[f].hi[d], P->ratio[d]);
			// a fine box must cover whole coarse cells (blocking_factor >= ratio): anything else is not conservative
			if (fine->boxes[f].lo[d] != cf.lo[d] * P->ratio[d] || fine->boxes[f].hi[d] != cf.hi[d] * P->ratio[d] + P->ratio[d] - 1) {
				delete P;
				return setError(ctx, QK_ERR_INVALID, "avgdown_plan_create: fine box not aligned with the refinement ratio");
			}
		}
		for (int c = 0; c < crse->nboxes; ++c) {
			AvgItem it{};
			it.crse_box = c;
			it.fine_box = f;
			bool ok = true;
			for (int d = 0; d < 3; ++d) {
				it.lo[d] = std::max(cf.lo[d], crse->boxes[c].lo[d]);
				it.hi[d] = std::min(cf.hi[d], crse->boxes[c].hi[d]);
				ok = ok && (it.lo[d] <= it.hi[d]);
			}
			if (ok) {
				P->items.push_back(it);
				P->max_cells = std::max<int64_t>(P->max_cells, static_cast<int64_t>(it.hi[0] - it.lo[0] + 1) * (it.hi[1] - it.lo[1] + 1) * (it.hi[2] - it.lo[2] + 1));
			}
		}
	}
	if (!P->items.empty() && ctx->device != QK_DEVICE_HOST_PLANNING) {
		if (hipMalloc(reinterpret_cast<void **>(&P->d_items), sizeof(AvgItem) * P->items.size()) != hipSuccess ||
		    hipMemcpy(P->d_items, P->items.data(), sizeof(AvgItem) * P->items.size(), hipMemcpyHostToDevice) != hipSuccess) {
			delete P;
			return setError(ctx, QK_ERR_HIP, "avgdown_plan_create: upload failed");
		}
	}
	*plan = P;
	return QK_OK;
}

int qk_avgdown_plan_destroy(qk_avgdown_plan *plan)
{
	if (plan != nullptr) {
		(void)hipFree(plan->d_items);
		delete plan;
	}
	return QK_OK;
}

int qk_avgdown_plan_num_items(qk_avgdown_plan *plan) { return plan == nullptr ? QK_ERR_INVALID : static_cast<int>(plan->items.size()); }

int qk_average_down(qk_avgdown_plan *plan, qk_stream s, const qk_array4 *fine_t, qk_array4 *crse_t, int scomp, int ncomp)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->crse->ctx;
	QK_REQUIRE(ctx, fine_t && crse_t && scomp >= 0 && ncomp >= 1, "average_down: bad argument");
	if (plan->items.empty()) {
		return QK_OK;
	}
	const dim3 grid(static_cast<unsigned>(std::min<int64_t>((plan->max_cells * ncomp + 255) / 256, 8192)), static_cast<unsigned>(plan->items.size()), 1);
	hipLaunchKernelGGL(k_average_down, grid, dim3(256), 0, static_cast<hipStream_t>(s), plan->d_items, fine_t, crse_t, scomp, ncomp, plan->ratio[0], plan->ratio[1],
			   plan->ratio[2]);
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------- box copy between arrays
namespace
{
__global__ void __launch_bounds__(256) k_copy_box(qk_array4 src, qk_array4 dst, int l0, int l1, int l2, int n0, int n1, int n2, int scomp, int dcomp, int ncomp)
{
	RA4 S(src);
	WA4 D(dst);
	const int64_t ncell = static_cast<int64_t>(n0) * n1 * n2;
	for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < ncell * ncomp; t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
		const int n = static_cast<int>(t / ncell);
		const int64_t c = t - n * ncell;
		const int k = static_cast<int>(c / (static_cast<int64_t>(n0) * n1));
		const int r = static_cast<int>(c - static_cast<int64_t>(k) * n0 * n1);
		const int j = r / n0;
		D(l0 + (r - j * n0), l1 + j, l2 + k, dcomp + n) = S(l0 + (r - j * n0), l1 + j, l2 + k, scomp + n);
	}
}
} // namespace

extern "C" int qk_copy_box(qk_ctx *ctx, qk_stream s, const qk_array4 *src, const qk_array4 *dst, const int lo[3], const int hi[3], int scomp, int dcomp, int ncomp)
{
	if (ctx == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(ctx, src && dst && lo && hi && ncomp >= 1, "copy_box: bad argument");
	const int n0 = hi[0] - lo[0] + 1, n1 = hi[1] - lo[1] + 1, n2 = hi[2] - lo[2] + 1;
	if (n0 <= 0 || n1 <= 0 || n2 <= 0) {
		return QK_OK;
	}
	const int64_t n = static_cast<int64_t>(n0) * n1 * n2 * ncomp;
	hipLaunchKernelGGL(k_copy_box, dim3(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 16384))), dim3(256), 0, static_cast<hipStream_t>(s), *src, *dst, lo[0],
			   lo[1], lo[2], n0, n1, n2, scomp, dcomp, ncomp);
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

// ---------------------------------------------------------------------------------------------- grid generation pieces
// amrex::AmrCore::MakeNewGrids turns tags into boxes with the Berger-Rigoutsos algorithm; AMReX is not vendored under
// /root/reference, so this repository uses a simpler rule with the same inputs (amr.n_error_buf, amr.blocking_factor,
// amr.max_grid_size): buffer the tags, refine every blocking-factor tile that holds a buffered tag, merge tiles greedily.
namespace
{
__global__ void __launch_bounds__(256) k_tags_to_tiles(const qk_box *boxes, const qk_carray4 *tags_t, int n_error_buf, int tile, int ndim, int n0, int n1, int n2,
						       int tx, int ty, int tz, int per0, int per1, int per2, int *flags)
{
	const int b = blockIdx.y;
	const qk_box bx = boxes[b];
	const int l0 = bx.hi[0] - bx.lo[0] + 1, l1 = bx.hi[1] - bx.lo[1] + 1, l2 = bx.hi[2] - bx.lo[2] + 1;
	const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (t >= static_cast<int64_t>(l0) * l1 * l2) {
		return;
	}
	const int k = static_cast<int>(t / (static_cast<int64_t>(l0) * l1));
	const int r = static_cast<int>(t - static_cast<int64_t>(k) * l0 * l1);
	const int j = r / l0;
	const int c[3] = {bx.lo[0] + (r - j * l0), bx.lo[1] + j, bx.lo[2] + k};
	A4<const char, qk_carray4> tag(tags_t[b]);
	if (tag(c[0], c[1], c[2]) != static_cast<char>(QK_TAG_SET)) {
		return;
	}
	// every tile that meets the cube of half-width n_error_buf around the tagged cell: clipped at a physical face of the domain, carried to
	// the other side through a periodic one (AMReX: TagBoxArray::mapPeriodicRemoveDuplicates)
	const int n[3] = {n0, n1, n2}, nt[3] = {tx, ty, tz}, per[3] = {per0, per1, per2};
	int a[3], e[3];
	for (int d = 0; d < 3; ++d) {
		const int nb = (d < ndim) ? n_error_buf : 0;
		int lo = c[d] - nb, hi = c[d] + nb;
		if (per[d] == 0 || d >= ndim) {
			lo = max(lo, 0);
			hi = min(hi, n[d] - 1);
		}
		// floor division: tile index of a cell beyond the lower face is negative
		a[d] = (d < ndim) ? ((lo >= 0) ? lo / tile : -((-lo + tile - 1) / tile)) : 0;
		e[d] = (d < ndim) ? hi / tile : 0;
		if (d < ndim && (per[d] == 0)) {
			e[d] = min(e[d], nt[d] - 1);
		}
	}
	for (int kk = a[2]; kk <= e[2]; ++kk) {
		for (int jj = a[1]; jj <= e[1]; ++jj) {
			for (int ii = a[0]; ii <= e[0]; ++ii) {
				const int wi = ((ii % tx) + tx) % tx, wj = ((jj % ty) + ty) % ty, wk = ((kk % tz) + tz) % tz;
				flags[wi + tx * (wj + ty * wk)] = 1; // benign race: every writer stores 1
			}
		}
	}
}
} // namespace

extern "C" {

int qk_amr_tile_flags(qk_level *lev, qk_stream s, const qk_carray4 *tags_t, const qk_box *domain, int n_error_buf, int tile, int *tile_flags_host)
{
	const int none[3] = {0, 0, 0};
	return qk_amr_tile_flags_periodic(lev, s, tags_t, domain, none, n_error_buf, tile, tile_flags_host);
}

int qk_amr_tile_flags_periodic(qk_level *lev, qk_stream s, const qk_carray4 *tags_t, const qk_box *domain, const int periodic[3], int n_error_buf, int tile,
			       int *tile_flags_host)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = lev->ctx;
	QK_REQUIRE(ctx, tags_t && domain && periodic && tile_flags_host && tile >= 1 && n_error_buf >= 0, "amr_tile_flags: bad argument");
	int n[3], nt[3];
	for (int d = 0; d < 3; ++d) {
		QK_REQUIRE(ctx, domain->lo[d] == 0, "amr_tile_flags: the domain must start at index 0");
		n[d] = domain->hi[d] + 1;
		nt[d] = (d < lev->ndim) ? (n[d] + tile - 1) / tile : 1;
	}
	const size_t bytes = sizeof(int) * static_cast<size_t>(nt[0]) * nt[1] * nt[2];
	if (lev->tile_flags_bytes < bytes) { // (kept on the level: a hierarchy regrids every other step, and hipMalloc / hipFree drain the device)
		(void)hipFree(lev->d_tile_flags);
		lev->d_tile_flags = nullptr;
		lev->tile_flags_bytes = 0;
		QK_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void **>(&lev->d_tile_flags), bytes));
		lev->tile_flags_bytes = bytes;
	}
	int *d_flags = lev->d_tile_flags;
	auto st = static_cast<hipStream_t>(s);
	int rc = QK_OK;
	if (hipMemsetAsync(d_flags, 0, bytes, st) != hipSuccess) {
		rc = setError(ctx, QK_ERR_HIP, "amr_tile_flags: memset failed");
	}
	if (rc == QK_OK && lev->nboxes > 0) {
		int64_t maxcells = 1;
		for (int d = 0; d < 3; ++d) {
			maxcells *= lev->maxlen[d];
		}
		const dim3 grid(static_cast<unsigned>((maxcells + 255) / 256), static_cast<unsigned>(lev->nboxes), 1);
		hipLaunchKernelGGL(k_tags_to_tiles, grid, dim3(256), 0, st, lev->d_boxes, tags_t, n_error_buf, tile, lev->ndim, n[0], n[1], n[2], nt[0], nt[1], nt[2], periodic[0], periodic[1],
				   periodic[2], d_flags);
	}
	if (rc == QK_OK) {
		if (hipGetLastError() != hipSuccess || hipMemcpyAsync(tile_flags_host, d_flags, bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
		    hipStreamSynchronize(st) != hipSuccess) {
			rc = setError(ctx, QK_ERR_HIP, "amr_tile_flags: kernel or copy failed");
		}
	}
	return rc;
}

// tiles[k][j][i] != 0 (tile = blocking_factor fine cells) -> fine boxes, merged greedily along x, then y, then z up to max_grid_size
// and never across a multiple of `parent_align` fine cells (0: no such restriction).  Returns the number of boxes (<= max_boxes)
// or a negative status; boxes are written in fine index space.
int qk_amr_cluster_tiles(const int *tiles, const int ntiles[3], int ndim, int blocking_factor, int max_grid_size, int parent_align, qk_box *boxes, int max_boxes)
{
	if (tiles == nullptr || ntiles == nullptr || boxes == nullptr || blocking_factor < 1 || max_grid_size < blocking_factor) {
		return QK_ERR_INVALID;
	}
	const int tx = ntiles[0], ty = ntiles[1], tz = ntiles[2];
	int maxt[3];
	for (int d = 0; d < 3; ++d) {
		maxt[d] = (d < ndim) ? std::max(max_grid_size / blocking_factor, 1) : 1;
	}
	const int align = (parent_align > 0) ? std::max(parent_align / blocking_factor, 1) : 0; // in tiles
	std::vector<char> used(static_cast<size_t>(tx) * ty * tz, 0);
	auto at = [&](int i, int j, int k) { return static_cast<size_t>(i) + static_cast<size_t>(tx) * (j + static_cast<size_t>(ty) * k); };
	auto free_tile = [&](int i, int j, int k) { return tiles[at(i, j, k)] != 0 && used[at(i, j, k)] == 0; };
	auto crosses = [&](int first, int next) { return align > 0 && (next / align) != (first / align); };
	int nb = 0;
	for (int k = 0; k < tz; ++k) {
		for (int j = 0; j < ty; ++j) {
			for (int i = 0; i < tx; ++i) {
				if (!free_tile(i, j, k)) {
					continue;
				}
				int i1 = i, j1 = j, k1 = k;
				while (i1 + 1 < tx && i1 + 1 - i < maxt[0] && !crosses(i, i1 + 1) && free_tile(i1 + 1, j, k)) {
					++i1;
				}
				auto row_free = [&](int jj, int kk) {
					for (int ii = i; ii <= i1; ++ii) {
						if (!free_tile(ii, jj, kk)) {
							return false;
						}
					}
					return true;
				};
				while (j1 + 1 < ty && j1 + 1 - j < maxt[1] && !crosses(j, j1 + 1) && row_free(j1 + 1, k)) {
					++j1;
				}
				auto plane_free = [&](int kk) {
					for (int jj = j; jj <= j1; ++jj) {
						if (!row_free(jj, kk)) {
							return false;
						}
					}
					return true;
				};
				while (k1 + 1 < tz && k1 + 1 - k < maxt[2] && !crosses(k, k1 + 1) && plane_free(k1 + 1)) {
					++k1;
				}
				for (int kk = k; kk <= k1; ++kk) {
					for (int jj = j; jj <= j1; ++jj) {
						for (int ii = i; ii <= i1; ++ii) {
							used[at(ii, jj, kk)] = 1;
						}
					}
				}
				if (nb >= max_boxes) {
					return QK_ERR_INVALID;
				}
				const int a[3] = {i, j, k}, e[3] = {i1, j1, k1};
				for (int d = 0; d < 3; ++d) {
					boxes[nb].lo[d] = (d < ndim) ? a[d] * blocking_factor : 0;
					boxes[nb].hi[d] = (d < ndim) ? (e[d] + 1) * blocking_factor - 1 : 0;
				}
				++nb;
			}
		}
	}
	return nb;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------- Berger-Rigoutsos clustering (host)
// amrex::AmrMesh::MakeNewGrids clusters the tagged cells, coarsened by blocking_factor / ref_ratio, with the point-clustering algorithm of
// Berger & Rigoutsos (1991) — amrex::ClusterList::chop(grid_eff) — refines the boxes back, merges what can be merged and cuts them to
// max_grid_size (reference tests/blast_amr_maxlev2.in:16-21 sets grid_eff = 0.7, blocking_factor = 32, n_error_buf = 3).  AMReX is an empty
// submodule under /root/reference; this restates the published algorithm with AMReX's documented choices (parity unpinned):
//   * a cluster = a set of tagged tiles with its bounding box; efficiency = tiles / volume of the box; clusters below grid_eff are cut in two;
//   * cut per direction from the signature (tagged tiles per plane): a hole (empty plane, the one nearest the middle) beats an inflection
//     (largest jump of the second difference across a sign change, at least 2 planes from either end, magnitude > 2) beats a bisection;
//     among the directions with the best kind of cut, the one whose cut leaves the longer shorter side (ties: the later direction);
//   * the boxes of all clusters are merged pairwise where two abut with equal cross-sections (BoxList::simplify) and cut into
//     ceil(len / max) nearly equal pieces per direction (BoxList::maxSize), everything in units of tiles, so every box edge is a multiple
//     of blocking_factor.
namespace
{
struct TPoint {
	int c[3];
};
struct TCluster {
	std::vector<TPoint> pts;
	int lo[3], hi[3];
	void minBox()
	{
		for (int d = 0; d < 3; ++d) {
			lo[d] = pts[0].c[d];
			hi[d] = pts[0].c[d];
		}
		for (auto const &p : pts) {
			for (int d = 0; d < 3; ++d) {
				lo[d] = std::min(lo[d], p.c[d]);
				hi[d] = std::max(hi[d], p.c[d]);
			}
		}
	}
	[[nodiscard]] auto eff() const -> double
	{
		double vol = 1.0;
		for (int d = 0; d < 3; ++d) {
			vol *= static_cast<double>(hi[d] - lo[d] + 1);
		}
		return static_cast<double>(pts.size()) / vol;
	}
};
enum CutStatus { HoleCut = 0, SteepCut, BisectCut, InvalidCut };

auto findCut(std::vector<int> const &hist, int lo, int hi, CutStatus &status) -> int
{
	const int MINOFF = 2, CUT_THRESH = 2;
	const int len = hi - lo + 1;
	status = InvalidCut;
	if (len <= 1) {
		return lo;
	}
	const int mid = len / 2;
	int cutpoint = -1;
	for (int i = 0; i < len; ++i) { // the empty plane nearest the middle
		if (hist[i] == 0) {
			status = HoleCut;
			if (std::abs(cutpoint - mid) > std::abs(i - mid)) {
				cutpoint = i;
				if (i > mid) {
					break;
				}
			}
		}
	}
	if (status == HoleCut) {
		return lo + cutpoint;
	}
	std::vector<int> dhist(len, 0);
	for (int i = 1; i < len - 1; ++i) {
		dhist[i] = hist[i + 1] - 2 * hist[i] + hist[i - 1];
	}
	int locmax = -1;
	for (int i = MINOFF; i < len - MINOFF; ++i) {
		const int iprev = dhist[i - 1], icur = dhist[i], locdif = std::abs(iprev - icur);
		if (iprev * icur < 0 && locdif >= locmax) {
			if (locdif > locmax) {
				status = SteepCut;
				cutpoint = i;
				locmax = locdif;
			} else if (std::abs(i - mid) < std::abs(cutpoint - mid)) {
				cutpoint = i;
			}
		}
	}
	if (locmax <= CUT_THRESH) {
		cutpoint = mid;
		status = BisectCut;
	}
	return lo + cutpoint;
}

// cut `cl` in two; the upper part is returned, the lower part stays (both with their minimal boxes)
auto chopCluster(TCluster &cl, int ndim) -> TCluster
{
	std::vector<int> hist[3];
	for (int d = 0; d < 3; ++d) {
		hist[d].assign(cl.hi[d] - cl.lo[d] + 1, 0);
	}
	for (auto const &p : cl.pts) {
		for (int d = 0; d < 3; ++d) {
			++hist[d][p.c[d] - cl.lo[d]];
		}
	}
	CutStatus status[3] = {InvalidCut, InvalidCut, InvalidCut}, mincut = InvalidCut;
	int cut[3] = {0, 0, 0};
	for (int d = 0; d < ndim; ++d) {
		cut[d] = findCut(hist[d], cl.lo[d], cl.hi[d], status[d]);
		if (status[d] < mincut) {
			mincut = status[d];
		}
	}
	int dir = -1, minlen = -1;
	for (int d = 0; d < ndim; ++d) {
		if (status[d] == mincut) {
			const int mincutlen = std::min(cut[d] - cl.lo[d], cl.hi[d] - cut[d]);
			if (mincutlen >= minlen) {
				dir = d;
				minlen = mincutlen;
			}
		}
	}
	TCluster up;
	if (dir < 0 || mincut == InvalidCut) {
		return up; // a single tile: cannot be cut (its efficiency is 1 anyway)
	}
	std::vector<TPoint> low;
	for (auto const &p : cl.pts) {
		(p.c[dir] < cut[dir] ? low : up.pts).push_back(p);
	}
	if (low.empty() || up.pts.empty()) { // (a hole cut at the first plane cannot happen for a minimal box; guard against an endless loop)
		up.pts.clear();
		return up;
	}
	cl.pts.swap(low);
	cl.minBox();
	up.minBox();
	return up;
}
} // namespace

extern "C" int qk_amr_cluster_berger_rigoutsos(const int *tiles, const int *allowed, const int ntiles[3], int ndim, int blocking_factor, int max_grid_size, double grid_eff,
					       qk_box *boxes, int max_boxes)
{
	if (tiles == nullptr || ntiles == nullptr || boxes == nullptr || blocking_factor < 1 || max_grid_size < blocking_factor || !(grid_eff > 0.0 && grid_eff <= 1.0)) {
		return QK_ERR_INVALID;
	}
	const int tx = ntiles[0], ty = ntiles[1], tz = ntiles[2];
	TCluster all;
	for (int k = 0; k < tz; ++k) {
		for (int j = 0; j < ty; ++j) {
			for (int i = 0; i < tx; ++i) {
				if (tiles[static_cast<size_t>(i) + static_cast<size_t>(tx) * (j + static_cast<size_t>(ty) * k)] != 0) {
					all.pts.push_back(TPoint{{i, j, k}});
				}
			}
		}
	}
	if (all.pts.empty()) {
		return 0;
	}
	all.minBox();
	// ClusterList::chop(eff): a cluster below the efficiency is cut; the part cut off goes to the end of the list, the rest is looked at again
	std::vector<TCluster> list;
	list.push_back(std::move(all));
	for (size_t n = 0; n < list.size();) {
		if (list[n].eff() < grid_eff) {
			TCluster up = chopCluster(list[n], ndim);
			if (up.pts.empty()) {
				++n;
			} else {
				list.push_back(std::move(up));
			}
		} else {
			++n;
		}
	}
	struct TB {
		int lo[3], hi[3];
	};
	std::vector<TB> bl;
	for (auto const &c : list) {
		TB b{};
		for (int d = 0; d < 3; ++d) {
			b.lo[d] = c.lo[d];
			b.hi[d] = c.hi[d];
		}
		bl.push_back(b);
	}
	// ClusterList::intersect(proper-nesting domain): a cluster box may hold unflagged tiles, and such a tile may lie where refinement is not
	// allowed.  A box with a forbidden tile is bisected along its longest edge until every piece is allowed; pieces without a flagged tile
	// are dropped (a flagged tile is always allowed, so this ends at single tiles at the latest).
	if (allowed != nullptr) {
		auto at = [&](int i, int j, int k) { return static_cast<size_t>(i) + static_cast<size_t>(tx) * (j + static_cast<size_t>(ty) * k); };
		std::vector<TB> keep, stack(bl.rbegin(), bl.rend());
		while (!stack.empty()) {
			const TB b = stack.back();
			stack.pop_back();
			bool anyFlag = false, allOk = true;
			for (int k = b.lo[2]; k <= b.hi[2]; ++k) {
				for (int j = b.lo[1]; j <= b.hi[1]; ++j) {
					for (int i = b.lo[0]; i <= b.hi[0]; ++i) {
						anyFlag = anyFlag || tiles[at(i, j, k)] != 0;
						allOk = allOk && allowed[at(i, j, k)] != 0;
					}
				}
			}
			if (!anyFlag) {
				continue;
			}
			if (allOk) {
				keep.push_back(b);
				continue;
			}
			int d = 0;
			for (int e = 1; e < 3; ++e) {
				if (b.hi[e] - b.lo[e] > b.hi[d] - b.lo[d]) {
					d = e;
				}
			}
			const int mid = (b.lo[d] + b.hi[d] + 1) / 2; // first tile of the upper half
			TB lower = b, upper = b;
			lower.hi[d] = mid - 1;
			upper.lo[d] = mid;
			stack.push_back(upper);
			stack.push_back(lower);
		}
		bl.swap(keep);
	}
	// BoxList::simplify: merge two boxes that abut along one direction with identical extents in the others, until nothing merges
	for (bool merged = true; merged;) {
		merged = false;
		for (size_t a = 0; a < bl.size() && !merged; ++a) {
			for (size_t b = a + 1; b < bl.size() && !merged; ++b) {
				for (int d = 0; d < ndim && !merged; ++d) {
					bool same = true;
					for (int e = 0; e < 3; ++e) {
						same = same && (e == d || (bl[a].lo[e] == bl[b].lo[e] && bl[a].hi[e] == bl[b].hi[e]));
					}
					if (same && (bl[a].hi[d] + 1 == bl[b].lo[d] || bl[b].hi[d] + 1 == bl[a].lo[d])) {
						bl[a].lo[d] = std::min(bl[a].lo[d], bl[b].lo[d]);
						bl[a].hi[d] = std::max(bl[a].hi[d], bl[b].hi[d]);
						bl.erase(bl.begin() + static_cast<long>(b));
						merged = true;
					}
				}
			}
		}
	}
	// BoxList::maxSize: ceil(len / max) pieces per direction, the first (len mod pieces) one tile longer
	const int maxt = std::max(max_grid_size / blocking_factor, 1);
	int nb = 0;
	for (auto const &b : bl) {
		int npc[3], bs[3], ex[3];
		for (int d = 0; d < 3; ++d) {
			const int len = b.hi[d] - b.lo[d] + 1;
			npc[d] = (d < ndim) ? (len + maxt - 1) / maxt : 1;
			bs[d] = len / npc[d];
			ex[d] = len - bs[d] * npc[d];
		}
		for (int kk = 0; kk < npc[2]; ++kk) {
			for (int jj = 0; jj < npc[1]; ++jj) {
				for (int ii = 0; ii < npc[0]; ++ii) {
					if (nb >= max_boxes) {
						return QK_ERR_INVALID;
					}
					const int idx[3] = {ii, jj, kk};
					for (int d = 0; d < 3; ++d) {
						const int first = b.lo[d] + idx[d] * bs[d] + std::min(idx[d], ex[d]);
						const int n = bs[d] + (idx[d] < ex[d] ? 1 : 0);
						boxes[nb].lo[d] = (d < ndim) ? first * blocking_factor : 0;
						boxes[nb].hi[d] = (d < ndim) ? (first + n) * blocking_factor - 1 : 0;
					}
					++nb;
				}
			}
		}
	}
	return nb;
}
