// qk_amr_fill.hip — coarse -> fine interpolation of the AMR level machinery:
//   the part of AMRSimulation::FillPatchWithData / amrex::FillPatchTwoLevels (reference src/simulation.hpp:1789-1858) that
//   fills fine cells no fine box covers from the time-interpolated coarse state, with
//     amrInterpMethod_ = 1: amrex::mf_linear_slope_minmax_interp (src/simulation.hpp:1389-1401) — conservative linear
//       interpolation, central slopes scaled by one factor per coarse cell and component so that no fine value leaves the
//       range of the 3x3x3 coarse neighbourhood;  amrInterpMethod_ = 0: piecewise constant (mf_pc_interp)
//     QuokkaSimulation::PreInterpState / PostInterpState (src/QuokkaSimulation.hpp:804-841) applied to the coarse stencil / the
//       new fine cell: the gas energy is interpolated as specific internal energy.
//   AMReX is not vendored under /root/reference: the interpolater is restated from its published description (AMReX
//   MFInterpolater docs) — parity with the reference is UNPINNED for this file; oracle/amr.hpp holds the same restatement.
// The plan (host) lists, per fine box, the boxes of ghost cells to fill: grown box minus every fine valid box (and its
// periodic images) minus cells beyond a non-periodic domain face, cut by the coarse boxes that provide the stencil.
#include <algorithm>
#include <vector>

#include "qk_device.hpp"
#include "qk_internal.hpp"

using namespace qk;

struct InterpItem {
	int fine_box, crse_box;
	int lo[3], hi[3]; // fine index space
};

struct qk_interp_plan {
	qk_level *crse = nullptr;
	qk_level *fine = nullptr;
	int ratio[3] = {2, 2, 2};
	std::vector<InterpItem> items;
	InterpItem *d_items = nullptr;
	int64_t max_cells = 0;
	int64_t max_ccells = 0; // coarse cells under the largest item
};

namespace
{

struct HBox {
	int lo[3], hi[3];
	[[nodiscard]] auto ok() const -> bool { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
};

auto isect(HBox const &a, HBox const &b) -> HBox
{
	HBox r{};
	for (int d = 0; d < 3; ++d) {
		r.lo[d] = std::max(a.lo[d], b.lo[d]);
		r.hi[d] = std::min(a.hi[d], b.hi[d]);
	}
	return r;
}

// a \ b as disjoint boxes
void boxDiff(HBox a, HBox const &b, std::vector<HBox> &out)
{
	HBox const c = isect(a, b);
	if (!c.ok()) {
		out.push_back(a);
		return;
	}
	for (int d = 0; d < 3; ++d) {
		if (a.lo[d] < c.lo[d]) {
			HBox p = a;
			p.hi[d] = c.lo[d] - 1;
			out.push_back(p);
			a.lo[d] = c.lo[d];
		}
		if (a.hi[d] > c.hi[d]) {
			HBox p = a;
			p.lo[d] = c.hi[d] + 1;
			out.push_back(p);
			a.hi[d] = c.hi[d];
		}
	}
}

auto floorDiv(int a, int r) -> int { return (a >= 0) ? a / r : -((-a + r - 1) / r); }

// One thread per (COARSE cell under the region, component): threadIdx.x runs over coarse cells, threadIdx.y over components.  The coarse value, its
// 27-point neighbourhood, the three limited slopes and the monotonicity factor alpha depend on the coarse cell alone; the r0 r1 r2 fine children take
// them with their own offsets.  (Round 3 had one thread per FINE cell loop over the components: 68 us per fill of a 64^3 box's ghost shell.)
// Round 5: the energy hook made the wave of the energy component do five times the work of the others — for each of its 27 neighbours the five
// hydro components (ten with time interpolation) and a division, 36 us per fill, the longest kernel of a small level's step.  The threads of the
// density, the momenta and the energy have just loaded exactly those values: they meet in LDS, plane by plane of the stencil, the specific internal
// energies of a plane's nine neighbours are computed by the block's component threads side by side, and the energy thread picks them up.  The same
// for PostInterpState: the children's densities and momenta go through LDS once, not once per child.  Same arithmetic per value: bit-identical.
constexpr int IT_CELLS = 64, IT_MAXCOMP = 16;
__global__ void __launch_bounds__(IT_CELLS *IT_MAXCOMP) k_interp(const InterpItem *items, qk_array4 *fine_t, const qk_array4 *crse_old_t, const qk_array4 *crse_new_t,
								 double w_old, double w_new, int ncomp, int method, int hooks, int ndim, int r0, int r1, int r2)
{
	__shared__ double s_nb[5][9][IT_CELLS]; // rho, px, py, pz, E of one stencil plane
	__shared__ double s_e[9][IT_CELLS];	// specific internal energy of that plane
	__shared__ double s_val[4][8][IT_CELLS]; // rho, px, py, pz of the (up to 8) children
	const InterpItem it = items[blockIdx.y];
	const int rr[3] = {r0, r1, r2};
	int clo[3];
	unsigned m[3];
	for (int d = 0; d < 3; ++d) {
		const int lo = (it.lo[d] >= 0) ? it.lo[d] / rr[d] : -((-it.lo[d] + rr[d] - 1) / rr[d]);
		const int hi = (it.hi[d] >= 0) ? it.hi[d] / rr[d] : -((-it.hi[d] + rr[d] - 1) / rr[d]);
		clo[d] = lo;
		m[d] = static_cast<unsigned>(hi - lo + 1);
	}
	const unsigned m01 = m[0] * m[1], ncc = m01 * m[2]; // (a piece of a ghost shell: far below 2^31 cells)
	WA4 F(fine_t[it.fine_box]);
	RA4 Co(crse_old_t[it.crse_box]);
	RA4 Cn(crse_new_t[it.crse_box]);
	const bool hk = (hooks != 0);
	const int tx = static_cast<int>(threadIdx.x), ty = static_cast<int>(threadIdx.y), ny = static_cast<int>(blockDim.y);
	// components in chunks of blockDim.y (<= IT_MAXCOMP; all of them at once up to 16 components: hydro 6, radiation-hydro 10; multigroup states take
	// several chunks — the hydro block, which the energy hook needs together, is in the first)
	for (int nbase = 0; nbase < ncomp; nbase += ny)
	for (unsigned base = blockIdx.x * IT_CELLS; base < ncc; base += gridDim.x * IT_CELLS) {
		const int n_raw = nbase + ty;
		const bool comp_live = n_raw < ncomp;
		const int n = comp_live ? n_raw : ncomp - 1;
		const bool hooked = hk && ncomp > ENE && nbase == 0; // (uniform) this chunk holds the hydro block and the energy hooks apply
		const unsigned t = base + threadIdx.x;
		const bool live = t < ncc;
		const unsigned tc = live ? t : 0;
		const unsigned kk = tc / m01;
		const unsigned r = tc - kk * m01;
		const unsigned jj = r / m[0];
		const int ic[3] = {clo[0] + static_cast<int>(r - jj * m[0]), clo[1] + static_cast<int>(jj), clo[2] + static_cast<int>(kk)};
		auto tv = [&](int i, int j, int k) -> double { // FillPatch time interpolation of this thread's component
			const double a = Co(i, j, k, n);
			if (w_new == 0.0) {
				return a;
			}
			return w_old * a + w_new * Cn(i, j, k, n);
		};
		// (every index into nb / val below is a compile-time constant after unrolling: dimensions a build does not have are masked, not looped
		// away — a run-time loop bound would put the arrays into scratch memory, which is what made this kernel take 33 us for 6500 cells)
		double nb[3][3][3];
		const bool lin = (method == 1);
#pragma unroll
		for (int c = -1; c <= 1; ++c) {
#pragma unroll
			for (int b = -1; b <= 1; ++b) {
#pragma unroll
				for (int a = -1; a <= 1; ++a) {
					const bool used = (a == 0 && b == 0 && c == 0) || (lin && (c == 0 || ndim == 3) && (b == 0 || ndim >= 2));
					nb[c + 1][b + 1][a + 1] = used ? tv(ic[0] + a, ic[1] + b, ic[2] + c) : 0.0;
				}
			}
		}
		if (hooked) { // PreInterpState on the stencil: the energy thread's values become (E - |p|^2 / (2 rho)) / rho
#pragma unroll
			for (int c = -1; c <= 1; ++c) {
				if (c != 0 && !(lin && ndim == 3)) { // (uniform)
					continue;
				}
				__syncthreads(); // (the previous plane's readers are done)
				if (n_raw <= ENE) {
#pragma unroll
					for (int b = -1; b <= 1; ++b) {
#pragma unroll
						for (int a = -1; a <= 1; ++a) {
							s_nb[n_raw][(b + 1) * 3 + (a + 1)][tx] = nb[c + 1][b + 1][a + 1];
						}
					}
				}
				__syncthreads();
				for (int q = ty; q < 9; q += ny) { // the plane's neighbours dealt to the block's component threads
					const int b = q / 3 - 1, a = q % 3 - 1;
					const bool used = (a == 0 && b == 0 && c == 0) || (lin && (b == 0 || ndim >= 2));
					if (used) {
						const double rho = s_nb[RHO][q][tx], px = s_nb[MX][q][tx], py = s_nb[MY][q][tx], pz = s_nb[MZ][q][tx], Etot = s_nb[ENE][q][tx];
						const double kinetic_energy = (px * px + py * py + pz * pz) / (2.0 * rho);
						s_e[q][tx] = (Etot - kinetic_energy) / rho;
					} else {
						s_e[q][tx] = 0.0;
					}
				}
				__syncthreads();
				if (n_raw == ENE) {
#pragma unroll
					for (int b = -1; b <= 1; ++b) {
#pragma unroll
						for (int a = -1; a <= 1; ++a) {
							nb[c + 1][b + 1][a + 1] = s_e[(b + 1) * 3 + (a + 1)][tx];
						}
					}
				}
			}
		}
		const double u = nb[1][1][1];
		double s[3] = {0., 0., 0.};
		double alpha = 1.0;
		if (method == 1) {
			double umax = u, umin = u;
#pragma unroll
			for (int c = -1; c <= 1; ++c) {
#pragma unroll
				for (int b = -1; b <= 1; ++b) {
#pragma unroll
					for (int a = -1; a <= 1; ++a) {
						if ((c == 0 || ndim == 3) && (b == 0 || ndim >= 2)) {
							const double v = nb[c + 1][b + 1][a + 1];
							umax = smax(umax, v);
							umin = smin(umin, v);
						}
					}
				}
			}
			s[0] = 0.5 * (nb[1][1][2] - nb[1][1][0]);
			if (ndim >= 2) {
				s[1] = 0.5 * (nb[1][2][1] - nb[1][0][1]);
			}
			if (ndim == 3) {
				s[2] = 0.5 * (nb[2][1][1] - nb[0][1][1]);
			}
			if (s[0] != 0.0 || s[1] != 0.0 || s[2] != 0.0) {
				const double dumax = fabs(s[0]) * static_cast<double>(rr[0] - 1) / (2.0 * rr[0]) + fabs(s[1]) * static_cast<double>(rr[1] - 1) / (2.0 * rr[1]) +
						     fabs(s[2]) * static_cast<double>(rr[2] - 1) / (2.0 * rr[2]);
				if (dumax * alpha > (umax - u)) {
					alpha = (umax - u) / dumax;
				}
				if (dumax * alpha > (u - umin)) {
					alpha = (u - umin) / dumax;
				}
			}
		}
		// the fine children of the coarse cell (at most 2 x 2 x 2)
		double val[8]; // child (cx, cy, cz) at index (cz * 2 + cy) * 2 + cx; children a ratio of 1 does not have are never stored
#pragma unroll
		for (int cz = 0; cz < 2; ++cz) {
#pragma unroll
			for (int cy = 0; cy < 2; ++cy) {
#pragma unroll
				for (int cx = 0; cx < 2; ++cx) {
					double v = u;
					if (method == 1) {
						const double off0 = (cx + 0.5) / r0 - 0.5, off1 = (cy + 0.5) / r1 - 0.5, off2 = (cz + 0.5) / r2 - 0.5;
						v = u + off0 * (s[0] * alpha) + off1 * (s[1] * alpha) + off2 * (s[2] * alpha);
					}
					val[(cz * 2 + cy) * 2 + cx] = v;
				}
			}
		}
		if (hooked) { // PostInterpState on the new fine cells: E = rho e + kinetic energy of the INTERPOLATED density and momenta
			__syncthreads(); // (the readers of the previous chunk of cells are done)
			if (n_raw < ENE) {
#pragma unroll
				for (int c = 0; c < 8; ++c) {
					s_val[n_raw][c][tx] = val[c];
				}
			}
			__syncthreads();
			if (n_raw == ENE) {
#pragma unroll
				for (int c = 0; c < 8; ++c) {
					const double rho = s_val[RHO][c][tx];
					const double px = s_val[MX][c][tx], py = s_val[MY][c][tx], pz = s_val[MZ][c][tx];
					const double Eint = rho * val[c];
					const double kinetic_energy = (px * px + py * py + pz * pz) / (2.0 * rho);
					val[c] = Eint + kinetic_energy;
				}
			}
		}
#pragma unroll
		for (int cz = 0; cz < 2; ++cz) {
#pragma unroll
			for (int cy = 0; cy < 2; ++cy) {
#pragma unroll
				for (int cx = 0; cx < 2; ++cx) {
					const int idx[3] = {ic[0] * r0 + cx, ic[1] * r1 + cy, ic[2] * r2 + cz};
					const bool inside = live && cx < r0 && cy < r1 && cz < r2 && idx[0] >= it.lo[0] && idx[0] <= it.hi[0] && idx[1] >= it.lo[1] &&
							    idx[1] <= it.hi[1] && idx[2] >= it.lo[2] && idx[2] <= it.hi[2];
					if (inside && comp_live) {
						F(idx[0], idx[1], idx[2], n) = val[(cz * 2 + cy) * 2 + cx];
					}
				}
			}
		}
	}
}

} // namespace

extern "C" {

int qk_interp_plan_create(qk_level *crse, qk_level *fine, const qk_geometry *fine_geom, int nghost, const int ratio[3], int whole_fab, int n_all_fine,
			  const qk_box *all_fine, qk_interp_plan **plan)
{
	if (crse == nullptr || fine == nullptr || plan == nullptr || ratio == nullptr || fine_geom == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = crse->ctx;
	QK_REQUIRE(ctx, crse->ctx == fine->ctx && crse->ndim == fine->ndim && nghost >= 0, "interp_plan_create: bad argument");
	for (int d = 0; d < fine->ndim; ++d) {
		QK_REQUIRE(ctx, ratio[d] == 1 || ratio[d] == 2, "interp_plan_create: refinement ratio must be 1 or 2 per direction");
	}
	auto *P = new qk_interp_plan;
	P->crse = crse;
	P->fine = fine;
	const int ndim = fine->ndim;
	for (int d = 0; d < 3; ++d) {
		P->ratio[d] = (d < ndim) ? ratio[d] : 1;
	}
	HBox dom{};
	int len[3];
	for (int d = 0; d < 3; ++d) {
		dom.lo[d] = fine_geom->domain.lo[d];
		dom.hi[d] = fine_geom->domain.hi[d];
		len[d] = dom.hi[d] - dom.lo[d] + 1;
	}
	// fine valid boxes and their periodic images
	std::vector<HBox> covered;
	int rng[3];
	for (int d = 0; d < 3; ++d) {
		rng[d] = (d < ndim && fine_geom->periodic[d] != 0) ? 1 : 0;
	}
	// the fine boxes of ALL ranks cover ghost cells (those are filled by the fine-fine exchange); default: this rank's boxes
	const int ncov = (all_fine != nullptr) ? n_all_fine : fine->nboxes;
	const qk_box *cov = (all_fine != nullptr) ? all_fine : fine->boxes.data();
	if (whole_fab == 0) {
		for (int b = 0; b < ncov; ++b) {
			for (int sz = -rng[2]; sz <= rng[2]; ++sz) {
				for (int sy = -rng[1]; sy <= rng[1]; ++sy) {
					for (int sx = -rng[0]; sx <= rng[0]; ++sx) {
						HBox v{};
						const int sh[3] = {sx * len[0], sy * len[1], sz * len[2]};
						for (int d = 0; d < 3; ++d) {
							v.lo[d] = cov[b].lo[d] + sh[d];
							v.hi[d] = cov[b].hi[d] + sh[d];
						}
						covered.push_back(v);
					}
				}
			}
		}
	}
	for (int b = 0; b < fine->nboxes; ++b) {
		HBox g{};
		for (int d = 0; d < 3; ++d) {
			const int ng = (d < ndim) ? nghost : 0;
			g.lo[d] = fine->boxes[b].lo[d] - ng;
			g.hi[d] = fine->boxes[b].hi[d] + ng;
			if (d < ndim && fine_geom->periodic[d] == 0) { // beyond a physical boundary: PhysBCFunct, not interpolation
				g.lo[d] = std::max(g.lo[d], dom.lo[d]);
				g.hi[d] = std::min(g.hi[d], dom.hi[d]);
			}
		}
		std::vector<HBox> todo{g};
		for (auto const &c : covered) {
			std::vector<HBox> next;
			for (auto const &t : todo) {
				boxDiff(t, c, next);
			}
			todo.swap(next);
		}
		// cut by the coarse boxes that hold the stencil: valid region first, then (periodic images) the ghost region
		for (int pass = 0; pass < 2 && !todo.empty(); ++pass) {
			for (int c = 0; c < crse->nboxes && !todo.empty(); ++c) {
				HBox rc{}; // fine cells whose coarse cell lies in the coarse valid box (pass 0) / within 2 ghost cells (pass 1)
				for (int d = 0; d < 3; ++d) {
					const int g2 = (pass == 1 && d < ndim) ? nghost / P->ratio[d] + ((nghost % P->ratio[d]) != 0 ? 1 : 0) : 0;
					rc.lo[d] = (crse->boxes[c].lo[d] - g2) * P->ratio[d];
					rc.hi[d] = (crse->boxes[c].hi[d] + g2) * P->ratio[d] + P->ratio[d] - 1;
				}
				std::vector<HBox> rest;
				for (auto const &t : todo) {
					HBox const piece = isect(t, rc);
					if (piece.ok()) {
						InterpItem it{};
						it.fine_box = b;
						it.crse_box = c;
						for (int d = 0; d < 3; ++d) {
							it.lo[d] = piece.lo[d];
							it.hi[d] = piece.hi[d];
						}
						P->items.push_back(it);
						P->max_cells = std::max<int64_t>(P->max_cells, static_cast<int64_t>(piece.hi[0] - piece.lo[0] + 1) * (piece.hi[1] - piece.lo[1] + 1) *
												       (piece.hi[2] - piece.lo[2] + 1));
						int64_t cc = 1;
						for (int d = 0; d < 3; ++d) {
							cc *= floorDiv(piece.hi[d], P->ratio[d]) - floorDiv(piece.lo[d], P->ratio[d]) + 1;
						}
						P->max_ccells = std::max(P->max_ccells, cc);
						boxDiff(t, rc, rest);
					} else {
						rest.push_back(t);
					}
				}
				todo.swap(rest);
			}
		}
		if (!todo.empty()) {
			delete P;
			return setError(ctx, QK_ERR_INVALID, "interp_plan_create: a fine ghost cell has no coarse cell underneath (level not properly nested)");
		}
	}
	if (!P->items.empty() && ctx->device != QK_DEVICE_HOST_PLANNING) {
		if (hipMalloc(reinterpret_cast<void **>(&P->d_items), sizeof(InterpItem) * P->items.size()) != hipSuccess ||
		    hipMemcpy(P->d_items, P->items.data(), sizeof(InterpItem) * P->items.size(), hipMemcpyHostToDevice) != hipSuccess) {
			delete P;
			return setError(ctx, QK_ERR_HIP, "interp_plan_create: upload failed");
		}
	}
	*plan = P;
	return QK_OK;
}

int qk_interp_plan_destroy(qk_interp_plan *plan)
{
	if (plan != nullptr) {
		(void)hipFree(plan->d_items);
		delete plan;
	}
	return QK_OK;
}

int qk_interp_plan_num_items(qk_interp_plan *plan) { return plan == nullptr ? QK_ERR_INVALID : static_cast<int>(plan->items.size()); }

int qk_interp_plan_item(qk_interp_plan *plan, int idx, int *fine_box, int *crse_box, int lo[3], int hi[3])
{
	if (plan == nullptr || idx < 0 || idx >= static_cast<int>(plan->items.size())) {
		return QK_ERR_INVALID;
	}
	auto const &it = plan->items[idx];
	*fine_box = it.fine_box;
	*crse_box = it.crse_box;
	for (int d = 0; d < 3; ++d) {
		lo[d] = it.lo[d];
		hi[d] = it.hi[d];
	}
	return QK_OK;
}

int qk_InterpFromCoarse(qk_interp_plan *plan, qk_stream s, qk_array4 *fine_t, const qk_array4 *crse_old_t, const qk_array4 *crse_new_t, double w_old,
			double w_new, int ncomp, int method, int energy_hooks)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->crse->ctx;
	QK_REQUIRE(ctx, fine_t && crse_old_t && crse_new_t && ncomp >= 1, "InterpFromCoarse: bad argument");
	QK_REQUIRE(ctx, method == 0 || method == 1, "InterpFromCoarse: interpolation method must be 0 (piecewise constant) or 1 (linear, min/max limited)");
	QK_REQUIRE(ctx, energy_hooks == 0 || ncomp > ENE, "InterpFromCoarse: the energy hooks need the hydro components");
	if (plan->items.empty()) {
		return QK_OK;
	}
	// a block covers IT_CELLS COARSE cells (round 4 sized the grid for the fine cells: seven of eight blocks returned at once)
	const dim3 grid(static_cast<unsigned>(std::min<int64_t>((plan->max_ccells + IT_CELLS - 1) / IT_CELLS, 16384)), static_cast<unsigned>(plan->items.size()), 1);
	hipLaunchKernelGGL(k_interp, grid, dim3(IT_CELLS, static_cast<unsigned>(std::min(ncomp, IT_MAXCOMP))), 0, static_cast<hipStream_t>(s), plan->d_items, fine_t, crse_old_t, crse_new_t, w_old, w_new, ncomp, method,
			   energy_hooks, plan->crse->ndim, plan->ratio[0], plan->ratio[1], plan->ratio[2]);
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

} // extern "C"
