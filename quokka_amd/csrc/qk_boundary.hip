// qk_boundary.hip — level-0 ghost fill (reference src/simulation.hpp:1751-1776):
//   state.FillBoundary(geom.periodicity())  ->  plan (host) + on-GPU copy kernel for same-rank neighbours
//                                               + pack / unpack kernels for strips that cross GPUs
//   PhysBCFunct(FilccCell + user functor)   ->  one kernel over the ghost shell
// The plan is pure host logic (box intersections with periodic shifts) and is the same on every rank;
// the wire order of a (sender, receiver) pair is (dst global box, src global box, shift) ascending on
// both sides, so the caller only has to move `send_count` doubles with RCCL p2p.
#include <algorithm>
#include <cstdlib>
#include <array>
#include <map>
#include <cstring>
#include <vector>

#include "qk_device.hpp"
#include "qk_internal.hpp"

using namespace qk;

namespace
{

struct CopyItem {
	int dst_box; // local index (receiver side) or -1
	int src_box; // local index (sender side) or -1
	int lo[3], hi[3]; // region in the DESTINATION index space
	int shift[3];	  // source index = destination index - shift
	int64_t offset;	  // offset (in doubles) into the peer buffer; unused for local copies
};

struct PeerPlan {
	int rank = -1;
	std::vector<CopyItem> send, recv;
	int64_t send_count = 0, recv_count = 0;
	CopyItem *d_send = nullptr, *d_recv = nullptr;
	int max_send_cells = 0, max_recv_cells = 0;
};

inline auto regionCells(CopyItem const &c) -> int64_t
{
	return static_cast<int64_t>(c.hi[0] - c.lo[0] + 1) * (c.hi[1] - c.lo[1] + 1) * (c.hi[2] - c.lo[2] + 1);
}

} // namespace

struct qk_ghost_plan {
	qk_level *lev = nullptr;
	qk_geometry geom{};
	int nghost = 0, ncomp = 0;
	std::vector<CopyItem> local;
	CopyItem *d_local = nullptr;
	int max_local_cells = 0;
	std::vector<PeerPlan> peers;
	// ghost slabs outside the domain in a non-periodic dimension (dst_box, lo, hi used)
	std::vector<CopyItem> shells; // slabs of boxes without remote-filled ghost cells first (n_shells_indep of them)
	CopyItem *d_shells = nullptr;
	int max_shell_cells = 0;
	int n_shells_indep = 0;
	std::vector<char> box_remote; // per local box: 1 if any ghost cell of the box is filled from another rank
	int active_scomp = 0, active_ncomp = -1; // component range of the same-rank copies and the physical BCs (-1: all)
	// device copy of the last BCRec / Dirichlet description (PhysBcArgs) and the host image it was made from: re-uploaded only when a
	// call brings different values (after a stream sync), so that the kernel reads them from global memory through a pointer
	void *d_physbc = nullptr;
	std::vector<unsigned char> h_physbc;
	qk_bcrec *d_bcs = nullptr;
	int d_bcs_n = 0;
	qk_dirichlet_face *d_dir = nullptr;
};

namespace
{

void partitionShells(qk_ghost_plan *P)
{
	auto indep = [&](CopyItem const &it) { return P->box_remote[it.dst_box] == 0; };
	std::stable_partition(P->shells.begin(), P->shells.end(), indep);
	P->n_shells_indep = static_cast<int>(std::count_if(P->shells.begin(), P->shells.end(), indep));
}

enum CopyMode { MODE_LOCAL = 0, MODE_PACK = 1, MODE_UNPACK = 2, MODE_SUM_LOCAL = 3, MODE_SUM_PACK = 4, MODE_SUM_UNPACK = 5 };

// blockIdx.y = item; grid-stride over region cells x ncomp
template <int MODE, typename T = double, typename D = qk_array4>
__global__ void __launch_bounds__(256) k_copy(const CopyItem *items, const D *state_t, T *buf, int ncomp, int scomp = 0)
{
	const CopyItem it = items[blockIdx.y];
	// 32-bit index arithmetic: a region holds fewer than 2^31 values (checked when the plan is made); a 64-bit division costs ~10x a 32-bit one
	// and there were four per element (the copy kernels were bound by them: round-3 profile, 56 -> us per launch at 256^3)
	const unsigned n0 = static_cast<unsigned>(it.hi[0] - it.lo[0] + 1), n1 = static_cast<unsigned>(it.hi[1] - it.lo[1] + 1),
		       n2 = static_cast<unsigned>(it.hi[2] - it.lo[2] + 1);
	const unsigned n01 = n0 * n1;
	const unsigned ncell = n01 * n2;
	const unsigned total = ncell * static_cast<unsigned>(ncomp);
	for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
		const unsigned un = t / ncell;
		const unsigned c = t - un * ncell;
		const unsigned uk = c / n01;
		const unsigned r = c - uk * n01;
		const unsigned uj = r / n0;
		const int n = static_cast<int>(un), k = static_cast<int>(uk), j = static_cast<int>(uj);
		const int i = static_cast<int>(r - uj * n0);
		const int di = it.lo[0] + i, dj = it.lo[1] + j, dk = it.lo[2] + k;
		if (MODE == MODE_LOCAL) {
			A4<T, D> Dst(state_t[it.dst_box]);
			A4<T, D> Src(state_t[it.src_box]);
			Dst(di, dj, dk, scomp + n) = Src(di - it.shift[0], dj - it.shift[1], dk - it.shift[2], scomp + n);
		} else if (MODE == MODE_PACK) {
			A4<T, D> Src(state_t[it.src_box]);
			buf[it.offset + t] = Src(di - it.shift[0], dj - it.shift[1], dk - it.shift[2], n);
		} else if (MODE == MODE_UNPACK) {
			A4<T, D> Dst(state_t[it.dst_box]);
			Dst(di, dj, dk, n) = buf[it.offset + t];
		} else if (MODE == MODE_SUM_LOCAL) { // SumBoundary: the ghost copy is added to the valid cell it mirrors
			A4<T, D> Dst(state_t[it.dst_box]);
			A4<T, D> Src(state_t[it.src_box]);
			atomicAdd(Src.ptr(di - it.shift[0], dj - it.shift[1], dk - it.shift[2], n), Dst(di, dj, dk, n));
		} else if (MODE == MODE_SUM_PACK) { // items = the strips this rank RECEIVES in FillBoundary: their ghost values go back
			A4<T, D> Dst(state_t[it.dst_box]);
			buf[it.offset + t] = Dst(di, dj, dk, n);
		} else { // items = the strips this rank SENDS in FillBoundary: add what came back to the valid cells
			A4<T, D> Src(state_t[it.src_box]);
			atomicAdd(Src.ptr(di - it.shift[0], dj - it.shift[1], dk - it.shift[2], n), buf[it.offset + t]);
		}
	}
}

// PhysBCFunct: FilccCell (AMReX_FilCC_3D_C.H) composed over dimensions + constant-Dirichlet user functor.
// The BCRecs and the Dirichlet model travel BY VALUE in the kernel arguments: the caller's arrays may be temporaries, and nothing of
// this call may depend on host memory after it returns.
constexpr int PHYSBC_MAX_COMP = QK_MAX_STATE_COMPS;
struct PhysBcArgs {
	qk_bcrec bcs[PHYSBC_MAX_COMP];
	qk_dirichlet_face dir[6];
	int has_dirichlet;
};

// HAS_DIR: a Dirichlet model is present (its own instantiation: the mathematical boundary types alone — reflecting walls, extrapolation — keep a
// short kernel; the functor code had made the common case 25 % slower)
template <bool HAS_DIR>
__global__ void __launch_bounds__(256) k_physbc(const CopyItem *items, qk_array4 *state_t, qk_geometry geom, int ncomp, const PhysBcArgs *pa, int scomp = 0)
{
	const qk_bcrec *bcs = pa->bcs;
	const qk_dirichlet_face *dirichlet = HAS_DIR ? pa->dir : nullptr;
	const CopyItem it = items[blockIdx.y];
	int lo[3], len[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		lo[d] = it.lo[d];
		len[d] = it.hi[d] - it.lo[d] + 1;
	}
	const unsigned l0 = static_cast<unsigned>(len[0]), l01 = l0 * static_cast<unsigned>(len[1]);
	const unsigned ncell = l01 * static_cast<unsigned>(len[2]); // (< 2^31: checked when the plan is made; 32-bit divisions, see k_copy)
	WA4 A(state_t[it.dst_box]);
	for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < ncell; t += gridDim.x * blockDim.x) {
		const unsigned uk = t / l01;
		const unsigned r = t - uk * l01;
		const unsigned uj = r / l0;
		const int k = static_cast<int>(uk), j = static_cast<int>(uj);
		const int i = static_cast<int>(r - uj * l0);
		const int idx[3] = {lo[0] + i, lo[1] + j, lo[2] + k};
		int side[3] = {0, 0, 0};
		bool out = false;
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			if (d < geom.ndim && geom.periodic[d] == 0) {
				if (idx[d] < geom.domain.lo[d]) {
					side[d] = -1;
					out = true;
				} else if (idx[d] > geom.domain.hi[d]) {
					side[d] = 1;
					out = true;
				}
			}
		}
		if (!out) {
			continue;
		}
		// user functor (closed set): constant state beyond an enabled face; x faces first, as a
		// setCustomBoundaryConditions written like HydroShocktube's (test_hydro_shocktube.cpp:94-144)
		const qk_dirichlet_face *df = nullptr;
		if constexpr (HAS_DIR) {
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				if (df == nullptr && side[d] != 0) {
					const qk_dirichlet_face *cand = &dirichlet[2 * d + (side[d] > 0 ? 1 : 0)];
					if (cand->enabled != 0) {
						df = cand;
					}
				}
			}
		}
		if (HAS_DIR && df != nullptr) {
			for (int n = scomp; n < scomp + ncomp; ++n) {
				A(idx[0], idx[1], idx[2], n) = df->values[n];
			}
			if (df->interior_mask != 0 || df->kinetic_from_interior != 0) {
				// RadTube's setCustomBoundaryConditions (test_radiation_tube.cpp:184-252)
				const int fd = static_cast<int>((df - dirichlet) / 2);
				const bool upper = ((df - dirichlet) & 1) != 0;
				int in[3] = {idx[0], idx[1], idx[2]};
				in[fd] = upper ? geom.domain.hi[fd] : geom.domain.lo[fd];
				for (int n = scomp; n < scomp + ncomp; ++n) {
					if (((df->interior_mask >> n) & 1ull) != 0) {
						A(idx[0], idx[1], idx[2], n) = A(in[0], in[1], in[2], n);
					}
				}
				if (df->kinetic_from_interior != 0 && ENE >= scomp && ENE < scomp + ncomp) {
					const double mom = A(in[0], in[1], in[2], MX + fd);
					const double Ekin = 0.5 * (mom * mom) / df->values[RHO];
					A(idx[0], idx[1], idx[2], ENE) = df->values[EINT] + Ekin;
				}
			}
			if (df->marshak != 0 && df->marshak_flux_comp >= scomp && df->marshak_flux_comp < scomp + ncomp) {
				// RadMarshak's setCustomBoundaryConditions (test_radiation_marshak.cpp:125-141), in its order of operations
				const int fd = static_cast<int>((df - dirichlet) / 2);
				int in[3] = {idx[0], idx[1], idx[2]};
				in[fd] = geom.domain.lo[fd];
				const double c = df->marshak_c;
				const double E_inc = df->values[df->marshak_energy_comp];
				const double E_0 = A(in[0], in[1], in[2], df->marshak_energy_comp);
				const double F_0 = A(in[0], in[1], in[2], df->marshak_flux_comp);
				A(idx[0], idx[1], idx[2], df->marshak_flux_comp) = 0.5 * c * E_inc - 0.5 * (c * E_0 + 2.0 * F_0);
			}
			continue;
		}
		for (int n = scomp; n < scomp + ncomp; ++n) {
			const qk_bcrec bc = bcs[n];
			int src[3] = {idx[0], idx[1], idx[2]};
			bool neg = false;
			bool any = false;
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				if (side[d] == 0) {
					continue;
				}
				const int type = (side[d] < 0) ? bc.lo[d] : bc.hi[d];
				const int edge = (side[d] < 0) ? geom.domain.lo[d] : geom.domain.hi[d];
				if (type == QK_BC_FOEXTRAP) {
					src[d] = edge;
					any = true;
				} else if (type == QK_BC_REFLECT_EVEN || type == QK_BC_REFLECT_ODD) {
					src[d] = (side[d] < 0) ? (2 * edge - idx[d] - 1) : (2 * edge - idx[d] + 1);
					neg = neg != (type == QK_BC_REFLECT_ODD);
					any = true;
				}
			}
			if (any) {
				const double v = A(src[0], src[1], src[2], n);
				A(idx[0], idx[1], idx[2], n) = neg ? -v : v;
			}
		}
	}
}

auto uploadItems(qk_ctx *ctx, std::vector<CopyItem> const &v, CopyItem **d) -> int
{
	*d = nullptr;
	if (v.empty() || ctx->device < 0) { // planning-only context keeps the plan on the host
		return QK_OK;
	}
	QK_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void **>(d), sizeof(CopyItem) * v.size()));
	QK_HIP_CHECK(ctx, hipMemcpy(*d, v.data(), sizeof(CopyItem) * v.size(), hipMemcpyHostToDevice));
	return QK_OK;
}

inline auto gridFor(int64_t cells, int nitems) -> dim3
{
	const int64_t gx = std::max<int64_t>(1, std::min<int64_t>((cells + 255) / 256, 256));
	return dim3(static_cast<unsigned>(gx), static_cast<unsigned>(nitems), 1);
}

} // namespace

extern "C" {

int qk_ghost_plan_create(qk_level *lev, qk_ghost_plan **plan_out, const qk_geometry *geom, int nghost, int ncomp, int n_all, const qk_box *all_boxes,
			 const int *owner, int my_rank)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = lev->ctx;
	QK_REQUIRE(ctx, plan_out && geom && all_boxes && owner && n_all > 0, "qk_ghost_plan_create: NULL argument");
	QK_REQUIRE(ctx, nghost >= 0 && ncomp > 0, "qk_ghost_plan_create: bad nghost/ncomp");
	// local index of every global box
	std::vector<int> local_of(n_all, -1);
	int nl = 0;
	for (int g = 0; g < n_all; ++g) {
		if (owner[g] == my_rank) {
			QK_REQUIRE(ctx, nl < lev->nboxes, "qk_ghost_plan_create: more owned boxes than the level holds");
			for (int d = 0; d < 3; ++d) {
				QK_REQUIRE(ctx, lev->boxes[nl].lo[d] == all_boxes[g].lo[d] && lev->boxes[nl].hi[d] == all_boxes[g].hi[d],
					   "qk_ghost_plan_create: owned boxes must be the level's boxes, in order");
			}
			local_of[g] = nl++;
		}
	}
	QK_REQUIRE(ctx, nl == lev->nboxes, "qk_ghost_plan_create: level has boxes this rank does not own");

	auto *P = new qk_ghost_plan;
	P->lev = lev;
	P->geom = *geom;
	P->nghost = nghost;
	P->ncomp = ncomp;

	// periodic shifts
	std::vector<std::array<int, 3>> shifts;
	int rng[3] = {0, 0, 0};
	for (int d = 0; d < geom->ndim; ++d) {
		rng[d] = (geom->periodic[d] != 0) ? 1 : 0;
	}
	for (int sz = -rng[2]; sz <= rng[2]; ++sz) {
		for (int sy = -rng[1]; sy <= rng[1]; ++sy) {
			for (int sx = -rng[0]; sx <= rng[0]; ++sx) {
				shifts.push_back({sx * (geom->domain.hi[0] - geom->domain.lo[0] + 1), sy * (geom->domain.hi[1] - geom->domain.lo[1] + 1),
						  sz * (geom->domain.hi[2] - geom->domain.lo[2] + 1)});
			}
		}
	}

	// QK_GHOST_LOOPBACK=1 (self-test of the production transport on ONE GPU): the pairs of this rank's own boxes are not items of the copy kernel
	// but strips for a peer whose rank is this rank's own — packed, sent to and received from itself (RCCL copies a grouped self send / recv on the
	// communication stream), unpacked: the stream ordering of pack -> send / recv -> unpack runs on hardware without a second GPU.
	const bool loopback = [] {
		const char *e = std::getenv("QK_GHOST_LOOPBACK");
		return e != nullptr && std::atoi(e) != 0;
	}();
	std::map<int, PeerPlan> peers;
	// canonical order: dst global box, src global box, shift
	for (int gd = 0; gd < n_all; ++gd) {
		qk_box grown = all_boxes[gd];
		for (int d = 0; d < geom->ndim; ++d) {
			grown.lo[d] -= nghost;
			grown.hi[d] += nghost;
		}
		for (int gs = 0; gs < n_all; ++gs) {
			const bool dst_mine = owner[gd] == my_rank;
			const bool src_mine = owner[gs] == my_rank;
			if (!dst_mine && !src_mine) {
				continue;
			}
			for (auto const &s : shifts) {
				if (gd == gs && s[0] == 0 && s[1] == 0 && s[2] == 0) {
					continue;
				}
				CopyItem it{};
				bool ok = true;
				for (int d = 0; d < 3; ++d) {
					it.lo[d] = std::max(grown.lo[d], all_boxes[gs].lo[d] + s[d]);
					it.hi[d] = std::min(grown.hi[d], all_boxes[gs].hi[d] + s[d]);
					it.shift[d] = s[d];
					ok = ok && (it.hi[d] >= it.lo[d]);
				}
				if (!ok) {
					continue;
				}
				it.dst_box = local_of[gd];
				it.src_box = local_of[gs];
				if (dst_mine && src_mine && loopback) {
					PeerPlan &pp = peers[my_rank]; // the same strip in the same place of both buffers
					pp.rank = my_rank;
					it.offset = pp.recv_count;
					pp.recv_count += regionCells(it) * ncomp;
					pp.send_count = pp.recv_count;
					pp.max_recv_cells = std::max<int64_t>(pp.max_recv_cells, regionCells(it));
					pp.max_send_cells = pp.max_recv_cells;
					pp.recv.push_back(it);
					pp.send.push_back(it);
				} else if (dst_mine && src_mine) {
					P->local.push_back(it);
					P->max_local_cells = std::max<int64_t>(P->max_local_cells, regionCells(it));
				} else if (dst_mine) {
					PeerPlan &pp = peers[owner[gs]];
					pp.rank = owner[gs];
					it.offset = pp.recv_count;
					pp.recv_count += regionCells(it) * ncomp;
					pp.max_recv_cells = std::max<int64_t>(pp.max_recv_cells, regionCells(it));
					pp.recv.push_back(it);
				} else {
					PeerPlan &pp = peers[owner[gd]];
					pp.rank = owner[gd];
					it.offset = pp.send_count;
					pp.send_count += regionCells(it) * ncomp;
					pp.max_send_cells = std::max<int64_t>(pp.max_send_cells, regionCells(it));
					pp.send.push_back(it);
				}
			}
		}
	}
	// ghost slabs beyond the domain faces (overlaps at edges/corners are written twice with the same value)
	for (int b = 0; b < lev->nboxes; ++b) {
		qk_box grown = lev->boxes[b];
		for (int d = 0; d < geom->ndim; ++d) {
			grown.lo[d] -= nghost;
			grown.hi[d] += nghost;
		}
		for (int d = 0; d < geom->ndim; ++d) {
			if (geom->periodic[d] != 0) {
				continue;
			}
			for (int side = 0; side < 2; ++side) {
				CopyItem it{};
				it.dst_box = b;
				it.src_box = b;
				for (int e = 0; e < 3; ++e) {
					it.lo[e] = grown.lo[e];
					it.hi[e] = grown.hi[e];
				}
				if (side == 0) {
					it.hi[d] = std::min(grown.hi[d], geom->domain.lo[d] - 1);
				} else {
					it.lo[d] = std::max(grown.lo[d], geom->domain.hi[d] + 1);
				}
				if (it.hi[d] >= it.lo[d]) {
					P->shells.push_back(it);
					P->max_shell_cells = std::max<int64_t>(P->max_shell_cells, regionCells(it));
				}
			}
		}
	}
	// boxes whose ghost cells depend on another rank; their physical-boundary slabs go last so that the slabs of the
	// independent boxes can be filled (and those boxes advanced) while the exchange is in flight
	P->box_remote.assign(lev->nboxes, 0);
	for (auto &kv : peers) {
		for (auto const &it : kv.second.recv) {
			P->box_remote[it.dst_box] = 1;
		}
	}
	partitionShells(P);
	{ // the copy kernels index a region with 32-bit arithmetic
		int64_t biggest = std::max<int64_t>(P->max_local_cells, P->max_shell_cells);
		for (auto &kv : peers) {
			biggest = std::max<int64_t>(biggest, std::max<int64_t>(kv.second.max_recv_cells, kv.second.max_send_cells));
		}
		if (biggest * ncomp >= (int64_t{1} << 31)) {
			delete P;
			return setError(ctx, QK_ERR_UNSUPPORTED, "qk_ghost_plan_create: a ghost region holds 2^31 or more values");
		}
	}
	int rc = uploadItems(ctx, P->local, &P->d_local);
	if (rc == QK_OK) {
		rc = uploadItems(ctx, P->shells, &P->d_shells);
	}
	for (auto &kv : peers) {
		if (rc == QK_OK) {
			rc = uploadItems(ctx, kv.second.send, &kv.second.d_send);
		}
		if (rc == QK_OK) {
			rc = uploadItems(ctx, kv.second.recv, &kv.second.d_recv);
		}
		P->peers.push_back(kv.second);
	}
	if (rc != QK_OK) {
		qk_ghost_plan_destroy(P);
		return rc;
	}
	*plan_out = P;
	return QK_OK;
}

int qk_ghost_plan_destroy(qk_ghost_plan *plan)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	(void)hipFree(plan->d_local);
	(void)hipFree(plan->d_shells);
	for (auto &p : plan->peers) {
		(void)hipFree(p.d_send);
		(void)hipFree(p.d_recv);
	}
	(void)hipFree(plan->d_bcs);
	(void)hipFree(plan->d_dir);
	(void)hipFree(plan->d_physbc);
	delete plan;
	return QK_OK;
}

int qk_ghost_plan_num_peers(qk_ghost_plan *plan) { return plan == nullptr ? QK_ERR_INVALID : static_cast<int>(plan->peers.size()); }

int qk_ghost_plan_peer(qk_ghost_plan *plan, int k, int *rank, int64_t *send_count, int64_t *recv_count)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(plan->lev->ctx, k >= 0 && k < static_cast<int>(plan->peers.size()) && rank && send_count && recv_count, "qk_ghost_plan_peer: bad index");
	*rank = plan->peers[k].rank;
	*send_count = plan->peers[k].send_count;
	*recv_count = plan->peers[k].recv_count;
	return QK_OK;
}

// plan introspection (host): kind 0 = same-rank copies, 1 = sends to peer k, 2 = receives from peer k, 3 = physical-boundary slabs
static auto planItems(qk_ghost_plan *plan, int kind, int k) -> std::vector<CopyItem> const *
{
	if (kind == 0) {
		return &plan->local;
	}
	if (kind == 3) {
		return &plan->shells;
	}
	if (k < 0 || k >= static_cast<int>(plan->peers.size())) {
		return nullptr;
	}
	return (kind == 1) ? &plan->peers[k].send : (kind == 2) ? &plan->peers[k].recv : nullptr;
}

int qk_ghost_plan_num_items(qk_ghost_plan *plan, int kind, int k)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	auto const *v = planItems(plan, kind, k);
	return v == nullptr ? QK_ERR_INVALID : static_cast<int>(v->size());
}

int qk_ghost_plan_item(qk_ghost_plan *plan, int kind, int k, int idx, int *dst_box, int *src_box, int lo[3], int hi[3], int shift[3], int64_t *offset)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	auto const *v = planItems(plan, kind, k);
	QK_REQUIRE(plan->lev->ctx, v != nullptr && idx >= 0 && idx < static_cast<int>(v->size()) && dst_box && src_box && lo && hi && shift && offset,
		   "qk_ghost_plan_item: bad argument");
	CopyItem const &it = (*v)[idx];
	*dst_box = it.dst_box;
	*src_box = it.src_box;
	for (int d = 0; d < 3; ++d) {
		lo[d] = it.lo[d];
		hi[d] = it.hi[d];
		shift[d] = it.shift[d];
	}
	*offset = it.offset;
	return QK_OK;
}

int qk_FillBoundary_local(qk_ghost_plan *plan, qk_stream s, qk_array4 *state_t)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->lev->ctx;
	QK_REQUIRE(ctx, state_t, "FillBoundary_local: NULL state");
	if (plan->local.empty()) {
		return QK_OK;
	}
	const int nc = (plan->active_ncomp < 0) ? plan->ncomp : plan->active_ncomp;
	ProfScope ps(ctx, static_cast<hipStream_t>(s), "ghost_copy_local");
	hipLaunchKernelGGL(k_copy<MODE_LOCAL>, gridFor(static_cast<int64_t>(plan->max_local_cells) * nc, static_cast<int>(plan->local.size())), dim3(256), 0,
			   static_cast<hipStream_t>(s), plan->d_local, state_t, static_cast<double *>(nullptr), nc, plan->active_scomp);
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

// iMultiFab flavour (redoFlag.FillBoundary, reference src/QuokkaSimulation.hpp:1157)
int qk_FillBoundary_local_int(qk_ghost_plan *plan, qk_stream s, qk_iarray4 *state_t)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->lev->ctx;
	QK_REQUIRE(ctx, state_t, "FillBoundary_local_int: NULL state");
	if (plan->local.empty()) {
		return QK_OK;
	}
	hipLaunchKernelGGL((k_copy<MODE_LOCAL, int, qk_iarray4>),
			   gridFor(static_cast<int64_t>(plan->max_local_cells) * plan->ncomp, static_cast<int>(plan->local.size())), dim3(256), 0,
			   static_cast<hipStream_t>(s), plan->d_local, state_t, static_cast<int *>(nullptr), plan->ncomp);
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

// amrex::FabArray::SumBoundary: every ghost value is added to the valid cell it is a copy of (same plan as FillBoundary, roles swapped)
int qk_SumBoundary_local(qk_ghost_plan *plan, qk_stream s, qk_array4 *state_t)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->lev->ctx;
	QK_REQUIRE(ctx, state_t, "SumBoundary_local: NULL state");
	if (plan->local.empty()) {
		return QK_OK;
	}
	hipLaunchKernelGGL(k_copy<MODE_SUM_LOCAL>, gridFor(static_cast<int64_t>(plan->max_local_cells) * plan->ncomp, static_cast<int>(plan->local.size())), dim3(256), 0,
			   static_cast<hipStream_t>(s), plan->d_local, state_t, static_cast<double *>(nullptr), plan->ncomp);
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

int qk_SumBoundary_pack(qk_ghost_plan *plan, qk_stream s, int k, const qk_array4 *state_t, double *buf)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->lev->ctx;
	QK_REQUIRE(ctx, k >= 0 && k < static_cast<int>(plan->peers.size()) && state_t && buf, "SumBoundary_pack: bad argument");
	PeerPlan const &pp = plan->peers[k];
	if (pp.recv.empty()) {
		return QK_OK;
	}
	hipLaunchKernelGGL(k_copy<MODE_SUM_PACK>, gridFor(static_cast<int64_t>(pp.max_recv_cells) * plan->ncomp, static_cast<int>(pp.recv.size())), dim3(256), 0,
			   static_cast<hipStream_t>(s), pp.d_recv, const_cast<qk_array4 *>(state_t), buf, plan->ncomp);
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

int qk_SumBoundary_unpack(qk_ghost_plan *plan, qk_stream s, int k, qk_array4 *state_t, const double *buf)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->lev->ctx;
	QK_REQUIRE(ctx, k >= 0 && k < static_cast<int>(plan->peers.size()) && state_t && buf, "SumBoundary_unpack: bad argument");
	PeerPlan const &pp = plan->peers[k];
	if (pp.send.empty()) {
		return QK_OK;
	}
	hipLaunchKernelGGL(k_copy<MODE_SUM_UNPACK>, gridFor(static_cast<int64_t>(pp.max_send_cells) * plan->ncomp, static_cast<int>(pp.send.size())), dim3(256), 0,
			   static_cast<hipStream_t>(s), pp.d_send, state_t, const_cast<double *>(buf), plan->ncomp);
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

int qk_FillBoundary_pack_int(qk_ghost_plan *plan, qk_stream s, int k, const qk_iarray4 *state_t, int *sendbuf)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->lev->ctx;
	QK_REQUIRE(ctx, k >= 0 && k < static_cast<int>(plan->peers.size()) && state_t && sendbuf, "FillBoundary_pack_int: bad argument");
	PeerPlan const &pp = plan->peers[k];
	if (pp.send.empty()) {
		return QK_OK;
	}
	hipLaunchKernelGGL((k_copy<MODE_PACK, int, qk_iarray4>), gridFor(static_cast<int64_t>(pp.max_send_cells) * plan->ncomp, static_cast<int>(pp.send.size())),
			   dim3(256), 0, static_cast<hipStream_t>(s), pp.d_send, const_cast<qk_iarray4 *>(state_t), sendbuf, plan->ncomp);
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

int qk_FillBoundary_unpack_int(qk_ghost_plan *plan, qk_stream s, int k, qk_iarray4 *state_t, const int *recvbuf)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->lev->ctx;
	QK_REQUIRE(ctx, k >= 0 && k < static_cast<int>(plan->peers.size()) && state_t && recvbuf, "FillBoundary_unpack_int: bad argument");
	PeerPlan const &pp = plan->peers[k];
	if (pp.recv.empty()) {
		return QK_OK;
	}
	hipLaunchKernelGGL((k_copy<MODE_UNPACK, int, qk_iarray4>), gridFor(static_cast<int64_t>(pp.max_recv_cells) * plan->ncomp, static_cast<int>(pp.recv.size())),
			   dim3(256), 0, static_cast<hipStream_t>(s), pp.d_recv, state_t, const_cast<int *>(recvbuf), plan->ncomp);
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

int qk_FillBoundary_pack(qk_ghost_plan *plan, qk_stream s, int k, const qk_array4 *state_t, double *sendbuf)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->lev->ctx;
	QK_REQUIRE(ctx, k >= 0 && k < static_cast<int>(plan->peers.size()) && state_t && sendbuf, "FillBoundary_pack: bad argument");
	PeerPlan const &pp = plan->peers[k];
	if (pp.send.empty()) {
		return QK_OK;
	}
	hipLaunchKernelGGL(k_copy<MODE_PACK>, gridFor(static_cast<int64_t>(pp.max_send_cells) * plan->ncomp, static_cast<int>(pp.send.size())), dim3(256), 0,
			   static_cast<hipStream_t>(s), pp.d_send, const_cast<qk_array4 *>(state_t), sendbuf, plan->ncomp);
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

int qk_FillBoundary_unpack(qk_ghost_plan *plan, qk_stream s, int k, qk_array4 *state_t, const double *recvbuf)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->lev->ctx;
	QK_REQUIRE(ctx, k >= 0 && k < static_cast<int>(plan->peers.size()) && state_t && recvbuf, "FillBoundary_unpack: bad argument");
	PeerPlan const &pp = plan->peers[k];
	if (pp.recv.empty()) {
		return QK_OK;
	}
	hipLaunchKernelGGL(k_copy<MODE_UNPACK>, gridFor(static_cast<int64_t>(pp.max_recv_cells) * plan->ncomp, static_cast<int>(pp.recv.size())), dim3(256), 0,
			   static_cast<hipStream_t>(s), pp.d_recv, state_t, const_cast<double *>(recvbuf), plan->ncomp);
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

int qk_ghost_plan_box_is_remote(qk_ghost_plan *plan, int local_box)
{
	if (plan == nullptr || local_box < 0 || local_box >= plan->lev->nboxes) {
		return QK_ERR_INVALID;
	}
	return plan->box_remote[local_box];
}

int qk_ghost_plan_set_box_remote(qk_ghost_plan *plan, int local_box, int flag)
{
	if (plan == nullptr || local_box < 0 || local_box >= plan->lev->nboxes) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->lev->ctx;
	plan->box_remote[local_box] = (flag != 0) ? 1 : 0;
	partitionShells(plan);
	if (plan->d_shells != nullptr && !plan->shells.empty()) {
		QK_HIP_CHECK(ctx, hipMemcpy(plan->d_shells, plan->shells.data(), sizeof(CopyItem) * plan->shells.size(), hipMemcpyHostToDevice));
	}
	return QK_OK;
}

int qk_FillPhysicalBoundary(qk_ghost_plan *plan, qk_stream s, qk_array4 *state_t, const qk_bcrec *bcs, const qk_dirichlet_face *dirichlet)
{
	return qk_FillPhysicalBoundary_subset(plan, s, state_t, bcs, dirichlet, QK_BOXES_ALL);
}

// Component range of the following qk_FillBoundary_local / qk_FillPhysicalBoundary* calls (ncomp < 0: all components again).  The strips
// exchanged with other ranks always carry every component.  The radiation substeps only read the radiation components of the ghost cells.
int qk_ghost_plan_set_components(qk_ghost_plan *plan, int scomp, int ncomp)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(plan->lev->ctx, ncomp < 0 || (scomp >= 0 && ncomp >= 1 && scomp + ncomp <= plan->ncomp), "ghost_plan_set_components: bad range");
	plan->active_scomp = (ncomp < 0) ? 0 : scomp;
	plan->active_ncomp = ncomp;
	return QK_OK;
}

int qk_FillPhysicalBoundary_subset(qk_ghost_plan *plan, qk_stream s, qk_array4 *state_t, const qk_bcrec *bcs, const qk_dirichlet_face *dirichlet,
				   int which)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_level *lev = plan->lev;
	qk_ctx *ctx = lev->ctx;
	QK_REQUIRE(ctx, state_t && bcs, "FillPhysicalBoundary: NULL argument");
	bool allPeriodic = true;
	for (int d = 0; d < plan->geom.ndim; ++d) {
		allPeriodic = allPeriodic && (plan->geom.periodic[d] != 0);
	}
	if (allPeriodic) { // simulation.hpp:1757
		return QK_OK;
	}
	(void)lev;
	// BCRecs / Dirichlet model are tiny and go by value into the kernel arguments (an asynchronous upload from the caller's arrays would
	// read them after this call has returned — they may be temporaries)
	QK_REQUIRE(ctx, plan->ncomp <= PHYSBC_MAX_COMP, "FillPhysicalBoundary: more than QK_MAX_STATE_COMPS (48) components");
	PhysBcArgs pa{};
	for (int n = 0; n < plan->ncomp; ++n) {
		pa.bcs[n] = bcs[n];
	}
	pa.has_dirichlet = (dirichlet != nullptr) ? 1 : 0;
	if (dirichlet != nullptr) {
		for (int f = 0; f < 6; ++f) {
			pa.dir[f] = dirichlet[f];
			if (dirichlet[f].enabled != 0 && dirichlet[f].marshak != 0) {
				QK_REQUIRE(ctx, (f % 2) == 0, "FillPhysicalBoundary: the Marshak condition is defined for lower faces only");
				QK_REQUIRE(ctx,
					   dirichlet[f].marshak_energy_comp >= 0 && dirichlet[f].marshak_energy_comp < plan->ncomp &&
					       dirichlet[f].marshak_flux_comp >= 0 && dirichlet[f].marshak_flux_comp < plan->ncomp &&
					       dirichlet[f].marshak_flux_comp != dirichlet[f].marshak_energy_comp && dirichlet[f].marshak_c > 0.0,
					   "FillPhysicalBoundary: bad Marshak face description");
			}
		}
	}
	QK_REQUIRE(ctx, which == QK_BOXES_ALL || which == QK_BOXES_LOCAL_ONLY || which == QK_BOXES_REMOTE_DEPENDENT, "FillPhysicalBoundary: bad subset");
	const int nall = static_cast<int>(plan->shells.size());
	const int first = (which == QK_BOXES_REMOTE_DEPENDENT) ? plan->n_shells_indep : 0;
	const int count = (which == QK_BOXES_ALL) ? nall : (which == QK_BOXES_LOCAL_ONLY ? plan->n_shells_indep : nall - plan->n_shells_indep);
	if (count == 0) {
		return QK_OK;
	}
	// the description lives in plan-owned device memory; a call with different values waits for the kernels that may still read the old
	// ones, then replaces them synchronously (nothing depends on the caller's arrays after this call returns).  Passing the ~1.3 KB by
	// value made every thread index a private copy: 577 us per launch instead of 82 (profiles/round1/v6 vs v4).
	if (plan->d_physbc == nullptr) {
		QK_HIP_CHECK(ctx, hipMalloc(&plan->d_physbc, sizeof(PhysBcArgs)));
	}
	if (plan->h_physbc.size() != sizeof(PhysBcArgs) || std::memcmp(plan->h_physbc.data(), &pa, sizeof(PhysBcArgs)) != 0) {
		// stream-ordered on `s`, then waited for: the copy is behind every kernel of this plan already queued on `s` and ahead of the launch
		// below also when `s` is a non-blocking stream (a plan is used from one stream at a time, include/quokka_amd.h)
		QK_HIP_CHECK(ctx, hipMemcpyAsync(plan->d_physbc, &pa, sizeof(PhysBcArgs), hipMemcpyHostToDevice, static_cast<hipStream_t>(s)));
		QK_HIP_CHECK(ctx, hipStreamSynchronize(static_cast<hipStream_t>(s)));
		plan->h_physbc.assign(reinterpret_cast<const unsigned char *>(&pa), reinterpret_cast<const unsigned char *>(&pa) + sizeof(PhysBcArgs));
	}
	ProfScope ps(ctx, static_cast<hipStream_t>(s), "ghost_physbc");
	if (pa.has_dirichlet != 0) {
		hipLaunchKernelGGL(k_physbc<true>, gridFor(plan->max_shell_cells, count), dim3(256), 0, static_cast<hipStream_t>(s), plan->d_shells + first, state_t,
				   plan->geom, (plan->active_ncomp < 0) ? plan->ncomp : plan->active_ncomp, static_cast<const PhysBcArgs *>(plan->d_physbc),
				   plan->active_scomp);
	} else {
		hipLaunchKernelGGL(k_physbc<false>, gridFor(plan->max_shell_cells, count), dim3(256), 0, static_cast<hipStream_t>(s), plan->d_shells + first, state_t,
				   plan->geom, (plan->active_ncomp < 0) ? plan->ncomp : plan->active_ncomp, static_cast<const PhysBcArgs *>(plan->d_physbc),
				   plan->active_scomp);
	}
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

} // extern "C"
