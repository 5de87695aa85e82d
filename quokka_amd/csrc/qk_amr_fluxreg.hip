// qk_amr_fluxreg.hip — flux register between a coarse level and the next finer one:
//   amrex::YAFluxRegister as driven by AMRSimulation::incrementFluxRegisters (reference src/simulation.hpp:1345-1387: CrseAdd / FineAdd
//   with the level's dt and cell size) and timeStepWithSubcycling (:1308: Reflux(state_new_cc_[lev]) before AverageDownTo).
// An item is a one-cell-thick slab of coarse cells just OUTSIDE one face of a fine box, not covered by any fine box and
// inside the (periodic) domain.  For such a cell o and direction d the register accumulates
//     fine box on the + side of o:   + (dt_c/dx_c) F_c(face o+1/2)  -  sum_substeps (dt_f/dx_c) <F_f>
//     fine box on the - side of o:   - (dt_c/dx_c) F_c(face o-1/2)  +  sum_substeps (dt_f/dx_c) <F_f>
// (<F_f> = mean of the r^2 fine face fluxes covering the coarse face) and Reflux adds it to the coarse state: the coarse
// flux through a coarse-fine interface is replaced by the time- and area-averaged fine flux.  AMReX is not vendored under
// /root/reference; the arithmetic order (sum of fine faces with the first transverse index fastest, one multiplication by
// dt_f / (dx_f r_x r_y r_z)) is this repository's restatement — parity with the reference UNPINNED; tests/test_amr_ops_gpu.py holds the numpy restatement.
#include <algorithm>
#include <vector>

#include "qk_device.hpp"
#include "qk_internal.hpp"

using namespace qk;

struct FrItem {
	int dir, side; // side 0: low face of the fine box (cells below it), 1: high face
	int fine_box, crse_box;
	int lo[3], hi[3]; // coarse cells adjacent to the fine box (unwrapped indices)
	int shift[3];	  // actual coarse cell = index + shift (periodic wrap)
	int64_t offset;	  // into the register buffer (in cells; component stride = total_cells)
};

struct qk_fluxreg {
	qk_level *crse = nullptr;
	qk_level *fine = nullptr;
	int ratio[3] = {2, 2, 2};
	int ncomp = 0;
	std::vector<FrItem> items;
	FrItem *d_items = nullptr;
	int group_begin[7] = {0, 0, 0, 0, 0, 0, 0}; // items sorted by (dir, side)
	int64_t total_cells = 0;
	int64_t max_cells = 0;
	double *d_reg = nullptr;
	int state_comp0 = 0; // Reflux: register component n is state component state_comp0 + n (qk_fluxreg_set_state_component)
	double *d_saved = nullptr; // qk_fluxreg_save / qk_fluxreg_restore (retries of the level that is the FINE side of this register)
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

enum { FR_CRSE_ADD = 0, FR_FINE_ADD = 1, FR_REFLUX = 2 };

// blockIdx.y = item of one (dir, side) group
// One launch over the items of ALL (dir, side) groups (round 5: three to six launches of a few microseconds each per call were 6 % of a small
// level's step).  CrseAdd / FineAdd write one register slot per (item, cell, component): no two items share a slot.  Reflux adds to state cells:
// two items can meet in one cell only if they belong to different groups (a coarse cell with fine boxes on two of its faces) — those launches
// stay separate per group (frLaunch).
struct FrFlux {
	const qk_array4 *t[3];
	double fac[3];
};
template <int MODE>
__global__ void __launch_bounds__(256) k_fluxreg(const FrItem *items, double *reg, int64_t total_cells, int ncomp, FrFlux fx, qk_array4 *state_t, int r0, int r1, int r2,
						 int scomp)
{
	const FrItem it = items[blockIdx.y];
	const qk_array4 *flux_t = fx.t[it.dir];
	const double fac = fx.fac[it.dir];
	const int n0 = it.hi[0] - it.lo[0] + 1, n1 = it.hi[1] - it.lo[1] + 1, n2 = it.hi[2] - it.lo[2] + 1;
	const int64_t ncell = static_cast<int64_t>(n0) * n1 * n2;
	const int rr[3] = {r0, r1, r2};
	const int d = it.dir;
	for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < ncell * ncomp; t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
		const int n = static_cast<int>(t / ncell);
		const int64_t c = t - n * ncell;
		const int kk = static_cast<int>(c / (static_cast<int64_t>(n0) * n1));
		const int r = static_cast<int>(c - static_cast<int64_t>(kk) * n0 * n1);
		const int jj = r / n0;
		const int o[3] = {it.lo[0] + (r - jj * n0), it.lo[1] + jj, it.lo[2] + kk};
		double *slot = reg + static_cast<int64_t>(n) * total_cells + it.offset + c;
		if (MODE == FR_CRSE_ADD) {
			// the interface face of cell o: its high face (o + e_d) if the fine box is above (side 0), its low face (o) otherwise
			RA4 F(flux_t[it.crse_box]);
			int f[3] = {o[0] + it.shift[0], o[1] + it.shift[1], o[2] + it.shift[2]};
			if (it.side == 0) {
				f[d] += 1;
			}
			const double v = fac * F(f[0], f[1], f[2], n);
			*slot = (it.side == 0) ? (*slot + v) : (*slot - v);
		} else if (MODE == FR_FINE_ADD) {
			RA4 F(flux_t[it.fine_box]);
			// fine faces covering the coarse face: plane index in d, r x r block in the transverse directions
			int base[3];
			for (int e = 0; e < 3; ++e) {
				base[e] = o[e] * rr[e];
			}
			base[d] = (it.side == 0) ? (o[d] + 1) * rr[d] : o[d] * rr[d];
			const int e1 = (d + 1) % 3, e2 = (d + 2) % 3;
			const int a1 = (e1 < e2) ? e1 : e2, a2 = (e1 < e2) ? e2 : e1; // a1 = lower axis (fastest)
			double sum = 0.0;
			for (int q = 0; q < rr[a2]; ++q) {
				for (int p = 0; p < rr[a1]; ++p) {
					int f[3] = {base[0], base[1], base[2]};
					f[a1] += p;
					f[a2] += q;
					sum += F(f[0], f[1], f[2], n);
				}
			}
			const double v = fac * sum;
			*slot = (it.side == 0) ? (*slot - v) : (*slot + v);
		} else {
			WA4 U(state_t[it.crse_box]);
			U(o[0] + it.shift[0], o[1] + it.shift[1], o[2] + it.shift[2], scomp + n) += *slot;
		}
	}
}

} // namespace

// skip_missing: a register cell that no local coarse box holds belongs to another rank's coarse part (qk_fluxreg_create_crse_part)
static int fluxregCreate(qk_level *crse, qk_level *fine, const qk_geometry *crse_geom, const int ratio[3], int ncomp, int n_all_fine, const qk_box *all_fine,
			 int reg_nghost, bool skip_missing, qk_fluxreg **fr)
{
	if (crse == nullptr || fine == nullptr || fr == nullptr || ratio == nullptr || crse_geom == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = crse->ctx;
	QK_REQUIRE(ctx, crse->ctx == fine->ctx && crse->ndim == fine->ndim && ncomp >= 1, "fluxreg_create: bad argument");
	auto *P = new qk_fluxreg;
	P->crse = crse;
	P->fine = fine;
	P->ncomp = ncomp;
	const int ndim = crse->ndim;
	int len[3], rng[3];
	HBox dom{};
	for (int d = 0; d < 3; ++d) {
		P->ratio[d] = (d < ndim) ? ratio[d] : 1;
		dom.lo[d] = crse_geom->domain.lo[d];
		dom.hi[d] = crse_geom->domain.hi[d];
		len[d] = dom.hi[d] - dom.lo[d] + 1;
		rng[d] = (d < ndim && crse_geom->periodic[d] != 0) ? 1 : 0;
	}
	// coarsened fine boxes: the local ones own registers; the boxes of ALL ranks (and their periodic images) mask cells that are refined
	std::vector<HBox> cfine(fine->nboxes), covered;
	auto coarsenBox = [&](qk_box const &b) {
		HBox c{};
		for (int d = 0; d < 3; ++d) {
			c.lo[d] = floorDiv(b.lo[d], P->ratio[d]);
			c.hi[d] = floorDiv(b.hi[d], P->ratio[d]);
		}
		return c;
	};
	for (int b = 0; b < fine->nboxes; ++b) {
		cfine[b] = coarsenBox(fine->boxes[b]);
	}
	const int ncov = (all_fine != nullptr) ? n_all_fine : fine->nboxes;
	const qk_box *cov = (all_fine != nullptr) ? all_fine : fine->boxes.data();
	for (int b = 0; b < ncov; ++b) {
		HBox const cb = coarsenBox(cov[b]);
		for (int sz = -rng[2]; sz <= rng[2]; ++sz) {
			for (int sy = -rng[1]; sy <= rng[1]; ++sy) {
				for (int sx = -rng[0]; sx <= rng[0]; ++sx) {
					HBox v = cb;
					const int sh[3] = {sx * len[0], sy * len[1], sz * len[2]};
					for (int d = 0; d < 3; ++d) {
						v.lo[d] += sh[d];
						v.hi[d] += sh[d];
					}
					covered.push_back(v);
				}
			}
		}
	}
	auto addItem = [&](int d, int side, int b, int c, HBox const &piece, const int sh[3]) {
		FrItem it{};
		it.dir = d;
		it.side = side;
		it.fine_box = b;
		it.crse_box = c;
		for (int e = 0; e < 3; ++e) {
			it.lo[e] = piece.lo[e] - sh[e];
			it.hi[e] = piece.hi[e] - sh[e];
			it.shift[e] = sh[e];
		}
		it.offset = P->total_cells;
		const int64_t n = static_cast<int64_t>(piece.hi[0] - piece.lo[0] + 1) * (piece.hi[1] - piece.lo[1] + 1) * (piece.hi[2] - piece.lo[2] + 1);
		P->total_cells += n;
		P->max_cells = std::max(P->max_cells, n);
		P->items.push_back(it);
	};
	for (int d = 0; d < ndim; ++d) {
		for (int side = 0; side < 2; ++side) {
			P->group_begin[2 * d + side] = static_cast<int>(P->items.size());
			for (int b = 0; b < fine->nboxes; ++b) {
				HBox slab = cfine[b];
				slab.lo[d] = slab.hi[d] = (side == 0) ? cfine[b].lo[d] - 1 : cfine[b].hi[d] + 1;
				for (int e = 0; e < ndim; ++e) { // no reflux through a physical boundary
					if (crse_geom->periodic[e] == 0) {
						slab.lo[e] = std::max(slab.lo[e], dom.lo[e]);
						slab.hi[e] = std::min(slab.hi[e], dom.hi[e]);
					}
				}
				if (!slab.ok()) {
					continue;
				}
				std::vector<HBox> todo{slab};
				for (auto const &c : covered) {
					std::vector<HBox> next;
					for (auto const &t : todo) {
						boxDiff(t, c, next);
					}
					todo.swap(next);
				}
				// pass 0: the cell (wrapped into the domain if it lies beyond a periodic face) is a valid cell of a local coarse box
				std::vector<HBox> rest = todo;
				for (int sz = -rng[2]; sz <= rng[2]; ++sz) {
					for (int sy = -rng[1]; sy <= rng[1]; ++sy) {
						for (int sx = -rng[0]; sx <= rng[0]; ++sx) {
							const int sh[3] = {sx * len[0], sy * len[1], sz * len[2]};
							for (int c = 0; c < crse->nboxes; ++c) {
								HBox cb{};
								for (int e = 0; e < 3; ++e) {
									cb.lo[e] = crse->boxes[c].lo[e] - sh[e]; // the coarse box in the unwrapped frame of the slab
									cb.hi[e] = crse->boxes[c].hi[e] - sh[e];
								}
								std::vector<HBox> next;
								for (auto const &t : rest) {
									HBox const piece = isect(t, cb);
									if (piece.ok()) {
										HBox w = piece;
										for (int e = 0; e < 3; ++e) {
											w.lo[e] += sh[e];
											w.hi[e] += sh[e];
										}
										addItem(d, side, b, c, w, sh);
										boxDiff(t, cb, next);
									} else {
										next.push_back(t);
									}
								}
								rest.swap(next);
							}
						}
					}
				}
				// pass 1: owned by another rank — accumulate in a ghost cell of a local coarse box; SumBoundary carries it to the owner.  The box must
				// hold the FACE between the register cell and the fine box (CrseAdd reads the coarse flux there): the one whose valid region the cell
				// adjoins in direction d — the coarse cell on the other side of the face lies under the fine box, hence in a local coarse box.  (Until
				// round 6 any box with the cell in its ghost ring was taken: a cell in a CORNER of the first box's ring made CrseAdd read that box's flux
				// array one row beyond its end — RadBeam on four ranks, profiles/round6/dist1.)
				if (reg_nghost > 0) {
					const int zero[3] = {0, 0, 0};
					for (int c = 0; c < crse->nboxes && !rest.empty(); ++c) {
						HBox gb{};
						for (int e = 0; e < 3; ++e) {
							const int g = (e == d) ? reg_nghost : 0;
							gb.lo[e] = crse->boxes[c].lo[e] - g;
							gb.hi[e] = crse->boxes[c].hi[e] + g;
						}
						std::vector<HBox> next;
						for (auto const &t : rest) {
							HBox const piece = isect(t, gb);
							if (piece.ok()) {
								addItem(d, side, b, c, piece, zero);
								boxDiff(t, gb, next);
							} else {
								next.push_back(t);
							}
						}
						rest.swap(next);
					}
				}
				if (!rest.empty() && !skip_missing) { // (single rank too: a register cell that no coarse box holds means the fine level is not properly nested —
						     // dropping it silently cost the advection hierarchy its conservation at the periodic faces)
					delete P;
					return setError(ctx, QK_ERR_INVALID, "fluxreg_create: a register cell is neither in a local coarse box nor in its ghost region");
				}
			}
		}
	}
	for (int g = 2 * ndim; g <= 6; ++g) {
		P->group_begin[g] = static_cast<int>(P->items.size());
	}
	if (!P->items.empty() && ctx->device != QK_DEVICE_HOST_PLANNING) {
		const size_t rb = sizeof(double) * static_cast<size_t>(P->total_cells) * ncomp;
		if (hipMalloc(reinterpret_cast<void **>(&P->d_items), sizeof(FrItem) * P->items.size()) != hipSuccess ||
		    hipMemcpy(P->d_items, P->items.data(), sizeof(FrItem) * P->items.size(), hipMemcpyHostToDevice) != hipSuccess ||
		    hipMalloc(reinterpret_cast<void **>(&P->d_reg), rb) != hipSuccess || hipMemsetAsync(P->d_reg, 0, rb, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
			delete P;
			return setError(ctx, QK_ERR_HIP, "fluxreg_create: allocation failed");
		}
	}
	*fr = P;
	return QK_OK;
}

extern "C" {

int qk_fluxreg_create(qk_level *crse, qk_level *fine, const qk_geometry *crse_geom, const int ratio[3], int ncomp, int n_all_fine, const qk_box *all_fine,
		      int reg_nghost, qk_fluxreg **fr)
{
	return fluxregCreate(crse, fine, crse_geom, ratio, ncomp, n_all_fine, all_fine, reg_nghost, false, fr);
}

int qk_fluxreg_create_crse_part(qk_level *crse, qk_level *all_fine_level, const qk_geometry *crse_geom, const int ratio[3], int ncomp, qk_fluxreg **fr)
{
	return fluxregCreate(crse, all_fine_level, crse_geom, ratio, ncomp, 0, nullptr, 0, true, fr);
}

int qk_fluxreg_destroy(qk_fluxreg *fr)
{
	if (fr != nullptr) {
		(void)hipFree(fr->d_items);
		(void)hipFree(fr->d_reg);
		(void)hipFree(fr->d_saved);
		delete fr;
	}
	return QK_OK;
}

int qk_fluxreg_num_items(qk_fluxreg *fr) { return fr == nullptr ? QK_ERR_INVALID : static_cast<int>(fr->items.size()); }

int qk_fluxreg_item(qk_fluxreg *fr, int idx, int *dir, int *side, int *fine_box, int *crse_box, int lo[3], int hi[3], int shift[3])
{
	if (fr == nullptr || idx < 0 || idx >= static_cast<int>(fr->items.size())) {
		return QK_ERR_INVALID;
	}
	auto const &it = fr->items[idx];
	*dir = it.dir;
	*side = it.side;
	*fine_box = it.fine_box;
	*crse_box = it.crse_box;
	for (int d = 0; d < 3; ++d) {
		lo[d] = it.lo[d];
		hi[d] = it.hi[d];
		shift[d] = it.shift[d];
	}
	return QK_OK;
}

int qk_fluxreg_reset(qk_fluxreg *fr, qk_stream s)
{
	if (fr == nullptr) {
		return QK_ERR_INVALID;
	}
	if (fr->d_reg != nullptr) {
		QK_HIP_CHECK(fr->crse->ctx, hipMemsetAsync(fr->d_reg, 0, sizeof(double) * static_cast<size_t>(fr->total_cells) * fr->ncomp, static_cast<hipStream_t>(s)));
	}
	return QK_OK;
}

// amrex::Copy(originalFineData, fr_as_fine->getFineData()) before the retry loop of a level and the copy back at every retry
// (reference src/QuokkaSimulation.hpp:894-900, :926-928).  Coarse and fine contributions share one array here; the coarse ones do not
// change while the fine level advances, so saving and restoring the whole register is the same operation.
int qk_fluxreg_save(qk_fluxreg *fr, qk_stream s)
{
	if (fr == nullptr) {
		return QK_ERR_INVALID;
	}
	const size_t bytes = sizeof(double) * static_cast<size_t>(fr->total_cells) * fr->ncomp;
	if (fr->d_reg == nullptr || bytes == 0) {
		return QK_OK;
	}
	if (fr->d_saved == nullptr) {
		QK_HIP_CHECK(fr->crse->ctx, hipMalloc(reinterpret_cast<void **>(&fr->d_saved), bytes));
	}
	QK_HIP_CHECK(fr->crse->ctx, hipMemcpyAsync(fr->d_saved, fr->d_reg, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(s)));
	return QK_OK;
}

int qk_fluxreg_restore(qk_fluxreg *fr, qk_stream s)
{
	if (fr == nullptr) {
		return QK_ERR_INVALID;
	}
	const size_t bytes = sizeof(double) * static_cast<size_t>(fr->total_cells) * fr->ncomp;
	if (fr->d_reg == nullptr || bytes == 0) {
		return QK_OK;
	}
	if (fr->d_saved == nullptr) {
		return qk::setError(fr->crse->ctx, QK_ERR_INVALID, "qk_fluxreg_restore: nothing saved");
	}
	QK_HIP_CHECK(fr->crse->ctx, hipMemcpyAsync(fr->d_reg, fr->d_saved, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(s)));
	return QK_OK;
}

static int frLaunch(qk_fluxreg *fr, qk_stream s, int mode, const qk_array4 *const flux[3], qk_array4 *state, const double fac[3])
{
	qk_ctx *ctx = fr->crse->ctx;
	auto st = static_cast<hipStream_t>(s);
	FrFlux fx{};
	for (int d = 0; d < 3; ++d) {
		fx.t[d] = (flux != nullptr) ? flux[d] : nullptr;
		fx.fac[d] = fac[d];
	}
	const auto gridOf = [&](int count) {
		return dim3(static_cast<unsigned>(std::min<int64_t>((fr->max_cells * fr->ncomp + 255) / 256, 1024)), static_cast<unsigned>(count), 1);
	};
	if (mode != FR_REFLUX) { // one slot per (item, cell, component): every group in one launch
		const int count = fr->group_begin[6];
		if (count > 0) {
			if (mode == FR_CRSE_ADD) {
				hipLaunchKernelGGL(k_fluxreg<FR_CRSE_ADD>, gridOf(count), dim3(256), 0, st, fr->d_items, fr->d_reg, fr->total_cells, fr->ncomp, fx, nullptr, fr->ratio[0],
						   fr->ratio[1], fr->ratio[2], fr->state_comp0);
			} else {
				hipLaunchKernelGGL(k_fluxreg<FR_FINE_ADD>, gridOf(count), dim3(256), 0, st, fr->d_items, fr->d_reg, fr->total_cells, fr->ncomp, fx, nullptr, fr->ratio[0],
						   fr->ratio[1], fr->ratio[2], fr->state_comp0);
			}
		}
	} else {
		for (int g = 0; g < 6; ++g) { // (a state cell can take increments from several groups: one launch per group, in order)
			const int first = fr->group_begin[g], count = fr->group_begin[g + 1] - first;
			if (count > 0) {
				hipLaunchKernelGGL(k_fluxreg<FR_REFLUX>, gridOf(count), dim3(256), 0, st, fr->d_items + first, fr->d_reg, fr->total_cells, fr->ncomp, fx, state,
						   fr->ratio[0], fr->ratio[1], fr->ratio[2], fr->state_comp0);
			}
		}
	}
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

int qk_fluxreg_CrseAdd(qk_fluxreg *fr, qk_stream s, const qk_array4 *const flux[3], const double dx[3], double dt)
{
	if (fr == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(fr->crse->ctx, flux && dx, "fluxreg_CrseAdd: NULL argument");
	const double fac[3] = {dt / dx[0], dt / dx[1], dt / dx[2]};
	return frLaunch(fr, s, FR_CRSE_ADD, flux, nullptr, fac);
}

int qk_fluxreg_FineAdd(qk_fluxreg *fr, qk_stream s, const qk_array4 *const flux[3], const double dx_fine[3], double dt)
{
	if (fr == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(fr->crse->ctx, flux && dx_fine, "fluxreg_FineAdd: NULL argument");
	const double rvol = static_cast<double>(fr->ratio[0] * fr->ratio[1] * fr->ratio[2]);
	const double fac[3] = {dt / (dx_fine[0] * rvol), dt / (dx_fine[1] * rvol), dt / (dx_fine[2] * rvol)};
	return frLaunch(fr, s, FR_FINE_ADD, flux, nullptr, fac);
}

int qk_fluxreg_set_state_component(qk_fluxreg *fr, int comp0)
{
	if (fr == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(fr->crse->ctx, comp0 >= 0 && comp0 + fr->ncomp <= QK_MAX_STATE_COMPS, "fluxreg_set_state_component: component range");
	fr->state_comp0 = comp0;
	return QK_OK;
}

int qk_fluxreg_Reflux(qk_fluxreg *fr, qk_stream s, qk_array4 *crse_state)
{
	if (fr == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(fr->crse->ctx, crse_state, "fluxreg_Reflux: NULL state");
	const double fac[3] = {0, 0, 0};
	return frLaunch(fr, s, FR_REFLUX, nullptr, crse_state, fac);
}

} // extern "C"
