// qk_amr_pcopy.hip — amrex::FabArray::ParallelCopy / ParallelAdd between two box layouts with independent owners:
//   the data motion of an AMR hierarchy whose refined boxes are distributed over ALL ranks (AMReX hands every new / remade level a
//   DistributionMapping of its own, reference src/simulation.hpp:1421-1500, :1657-1702): coarse data under a fine box's ghost cells
//   (amrex::FillPatchTwoLevels fills a coarse patch on the FINE level's distribution by ParallelCopy, then interpolates locally,
//   reference src/simulation.hpp:1789-1858), averaged-down fine data on its way to the coarse owner (amrex::average_down builds the
//   coarsened fine MultiFab, then ParallelCopy; :1949-1964), the fine side of a flux register (amrex::YAFluxRegister::Reflux:
//   m_crse_data.ParallelAdd(m_cfpatch); :1308) and the old-level data a remade level keeps (:1672-1685).
// Same structure as the ghost plan (qk_boundary.hip): pure host box algebra, identical on every rank; same-rank pairs become items of
// one copy kernel, pairs that cross ranks are packed into one contiguous buffer per peer in the canonical order (destination global
// box, source global box, source piece, shift) and travel as one RCCL send / recv pair.
#include <algorithm>
#include <array>
#include <cstdlib>
#include <map>
#include <vector>

#include "qk_device.hpp"
#include "qk_internal.hpp"

using namespace qk;

namespace
{

struct PcItem {
	int dst_box; // local index on the receiving side, -1 otherwise
	int src_box; // local index on the sending side, -1 otherwise
	int lo[3], hi[3]; // region in the DESTINATION index space
	int shift[3];	  // source index = destination index - shift
	int64_t offset;	  // into the peer buffer (values)
};

struct PcPeer {
	int rank = -1;
	std::vector<PcItem> send, recv;
	std::vector<int> recv_groups; // (see addGroups)
	int64_t send_count = 0, recv_count = 0;
	int64_t max_send_cells = 0, max_recv_cells = 0;
	PcItem *d_send = nullptr, *d_recv = nullptr;
};

struct HB {
	int lo[3], hi[3];
	[[nodiscard]] auto ok() const -> bool { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
};

auto isect(HB const &a, HB const &b) -> HB
{
	HB r{};
	for (int d = 0; d < 3; ++d) {
		r.lo[d] = std::max(a.lo[d], b.lo[d]);
		r.hi[d] = std::min(a.hi[d], b.hi[d]);
	}
	return r;
}

// a \ b as disjoint boxes
void boxDiff(HB a, HB const &b, std::vector<HB> &out)
{
	HB const c = isect(a, b);
	if (!c.ok()) {
		out.push_back(a);
		return;
	}
	for (int d = 0; d < 3; ++d) {
		if (a.lo[d] < c.lo[d]) {
			HB p = a;
			p.hi[d] = c.lo[d] - 1;
			out.push_back(p);
			a.lo[d] = c.lo[d];
		}
		if (a.hi[d] > c.hi[d]) {
			HB p = a;
			p.lo[d] = c.hi[d] + 1;
			out.push_back(p);
			a.hi[d] = c.hi[d];
		}
	}
}

inline auto cells(PcItem const &c) -> int64_t { return static_cast<int64_t>(c.hi[0] - c.lo[0] + 1) * (c.hi[1] - c.lo[1] + 1) * (c.hi[2] - c.lo[2] + 1); }

enum { PC_LOCAL = 0, PC_PACK = 1, PC_UNPACK = 2 };

// blockIdx.y = item; grid-stride over region cells x ncomp (32-bit index arithmetic, as k_copy of the ghost plan).  ADD: the value is added to
// the destination.  Several source pieces may land on one destination cell (the rings of two or three fine boxes around one coarse cell): the
// items of a plan are sorted into groups whose destination regions are disjoint (addGroups) and an add is one launch per group, in order —
// plain read-add-write, the same sum in the same order on every run (atomics gave (s + a) + b or (s + b) + a as the scheduler pleased).
template <int MODE, bool ADD>
__global__ void __launch_bounds__(256) k_pcopy(const PcItem *items, const qk_array4 *src_t, qk_array4 *dst_t, double *buf, int ncomp, int scomp_src, int scomp_dst)
{
	const PcItem it = items[blockIdx.y];
	const unsigned n0 = static_cast<unsigned>(it.hi[0] - it.lo[0] + 1), n1 = static_cast<unsigned>(it.hi[1] - it.lo[1] + 1),
		       n2 = static_cast<unsigned>(it.hi[2] - it.lo[2] + 1);
	const unsigned n01 = n0 * n1, ncell = n01 * n2, total = ncell * static_cast<unsigned>(ncomp);
	for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
		const unsigned un = t / ncell;
		const unsigned c = t - un * ncell;
		const unsigned uk = c / n01;
		const unsigned r = c - uk * n01;
		const unsigned uj = r / n0;
		const int n = static_cast<int>(un);
		const int di = it.lo[0] + static_cast<int>(r - uj * n0), dj = it.lo[1] + static_cast<int>(uj), dk = it.lo[2] + static_cast<int>(uk);
		double v;
		if (MODE == PC_UNPACK) {
			v = buf[it.offset + t];
		} else {
			RA4 S(src_t[it.src_box]);
			v = S(di - it.shift[0], dj - it.shift[1], dk - it.shift[2], scomp_src + n);
		}
		if (MODE == PC_PACK) {
			buf[it.offset + t] = v;
		} else {
			WA4 D(dst_t[it.dst_box]);
			if (ADD) {
				double *d = D.ptr(di, dj, dk, scomp_dst + n);
				*d = *d + v;
			} else {
				D(di, dj, dk, scomp_dst + n) = v;
			}
		}
	}
}

inline auto gridFor(int64_t values, int nitems) -> dim3
{
	const int64_t gx = std::max<int64_t>(1, std::min<int64_t>((values + 255) / 256, 256));
	return dim3(static_cast<unsigned>(gx), static_cast<unsigned>(nitems), 1);
}
constexpr int PC_MAX_ITEMS_PER_LAUNCH = 65535; // (the items of a launch live in gridDim.y)

// Stable-sorts `items` into groups in which no two items touch the same destination cell (greedy colouring in the canonical item order: an item takes
// the first group none of whose members overlaps it) and returns the group boundaries [0, g1, g2, ..., n].  Host box algebra, identical on every rank.
auto addGroups(std::vector<PcItem> &items) -> std::vector<int>
{
	const int n = static_cast<int>(items.size());
	std::vector<int> colour(n, 0);
	int ncol = (n > 0) ? 1 : 0;
	for (int i = 0; i < n; ++i) {
		std::vector<char> used(static_cast<size_t>(ncol) + 1, 0);
		for (int j = 0; j < i; ++j) {
			if (items[j].dst_box != items[i].dst_box) {
				continue;
			}
			bool hit = true;
			for (int d = 0; d < 3; ++d) {
				hit = hit && items[j].lo[d] <= items[i].hi[d] && items[i].lo[d] <= items[j].hi[d];
			}
			if (hit) {
				used[colour[j]] = 1;
			}
		}
		int c = 0;
		while (used[c] != 0) {
			++c;
		}
		colour[i] = c;
		ncol = std::max(ncol, c + 1);
	}
	std::vector<int> order(n);
	for (int i = 0; i < n; ++i) {
		order[i] = i;
	}
	std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return colour[a] < colour[b]; });
	std::vector<PcItem> sorted(n);
	std::vector<int> bounds{0};
	for (int q = 0; q < n; ++q) {
		sorted[q] = items[order[q]];
		if (q > 0 && colour[order[q]] != colour[order[q - 1]]) {
			bounds.push_back(q);
		}
	}
	bounds.push_back(n);
	items.swap(sorted);
	return bounds;
}

// one launch per run of at most 65535 items of [first, last)
template <class Launch> void forItemChunks(int first, int last, Launch &&launch)
{
	for (int a = first; a < last; a += PC_MAX_ITEMS_PER_LAUNCH) {
		launch(a, std::min(last - a, PC_MAX_ITEMS_PER_LAUNCH));
	}
}

} // namespace

struct qk_pcopy_plan {
	qk_ctx *ctx = nullptr;
	int ncomp = 0; // values per cell in the peer buffers (fixed at creation, like the ghost plan's)
	int nsrc_local = 0, ndst_local = 0;
	std::vector<PcItem> local;
	std::vector<int> local_groups; // add groups of `local` (addGroups)
	PcItem *d_local = nullptr;
	int64_t max_local_cells = 0;
	std::vector<PcPeer> peers;
};

extern "C" {

int qk_pcopy_plan_create(qk_ctx *ctx, const qk_geometry *geom, int n_src, const qk_box *src_boxes, const int *src_owner, int src_nghost, int src_ring_only, int n_dst,
			 const qk_box *dst_boxes, const int *dst_owner, int dst_nghost, const qk_box *dst_holes, int ncomp, int my_rank, qk_pcopy_plan **plan_out)
{
	if (ctx == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(ctx, geom && plan_out && n_src >= 0 && n_dst >= 0 && (n_src == 0 || (src_boxes && src_owner)) && (n_dst == 0 || (dst_boxes && dst_owner)),
		   "qk_pcopy_plan_create: NULL argument");
	QK_REQUIRE(ctx, src_nghost >= 0 && dst_nghost >= 0 && (src_ring_only == 0 || src_nghost > 0), "qk_pcopy_plan_create: bad ghost widths");
	QK_REQUIRE(ctx, ncomp >= 1 && ncomp <= QK_MAX_STATE_COMPS, "qk_pcopy_plan_create: bad component count");
	auto *P = new qk_pcopy_plan;
	P->ctx = ctx;
	P->ncomp = ncomp;
	const int ndim = geom->ndim;
	std::vector<int> src_local(n_src, -1), dst_local(n_dst, -1);
	for (int g = 0; g < n_src; ++g) {
		if (src_owner[g] == my_rank) {
			src_local[g] = P->nsrc_local++;
		}
	}
	for (int g = 0; g < n_dst; ++g) {
		if (dst_owner[g] == my_rank) {
			dst_local[g] = P->ndst_local++;
		}
	}
	std::vector<std::array<int, 3>> shifts;
	int rng[3] = {0, 0, 0};
	for (int d = 0; d < ndim; ++d) {
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
	auto grown = [&](qk_box const &b, int ng) {
		HB g{};
		for (int d = 0; d < 3; ++d) {
			const int w = (d < ndim) ? ng : 0;
			g.lo[d] = b.lo[d] - w;
			g.hi[d] = b.hi[d] + w;
		}
		return g;
	};
	// source pieces per source box: the (grown) box, or its ghost ring alone
	std::vector<std::vector<HB>> src_pieces(n_src);
	for (int gs = 0; gs < n_src; ++gs) {
		HB const g = grown(src_boxes[gs], src_nghost);
		if (src_ring_only != 0) {
			boxDiff(g, grown(src_boxes[gs], 0), src_pieces[gs]);
		} else {
			src_pieces[gs].push_back(g);
		}
	}
	std::map<int, PcPeer> peers;
	int64_t biggest = 0;
	// QK_GHOST_LOOPBACK=1 (as in qk_ghost_plan_create: the production transport on ONE GPU): same-rank pairs are not items of the copy kernel but regions
	// packed for / unpacked from a peer whose rank is this rank's own — the stream ordering of pack -> send / recv -> unpack of a ParallelCopy / ParallelAdd
	// runs on hardware without a second GPU
	const bool loopback = [] {
		const char *e = std::getenv("QK_GHOST_LOOPBACK");
		return e != nullptr && std::atoi(e) != 0;
	}();
	for (int gd = 0; gd < n_dst; ++gd) {
		std::vector<HB> dst_pieces;
		HB const g = grown(dst_boxes[gd], dst_nghost);
		if (dst_holes != nullptr) {
			boxDiff(g, grown(dst_holes[gd], 0), dst_pieces);
		} else {
			dst_pieces.push_back(g);
		}
		const bool dst_mine = dst_owner[gd] == my_rank;
		for (int gs = 0; gs < n_src; ++gs) {
			const bool src_mine = src_owner[gs] == my_rank;
			if (!dst_mine && !src_mine) {
				continue;
			}
			for (auto const &sp : src_pieces[gs]) {
				for (auto const &s : shifts) {
					HB shifted = sp;
					for (int d = 0; d < 3; ++d) {
						shifted.lo[d] += s[d];
						shifted.hi[d] += s[d];
					}
					for (auto const &dp : dst_pieces) {
						HB const r = isect(dp, shifted);
						if (!r.ok()) {
							continue;
						}
						PcItem it{};
						for (int d = 0; d < 3; ++d) {
							it.lo[d] = r.lo[d];
							it.hi[d] = r.hi[d];
							it.shift[d] = s[d];
						}
						it.dst_box = dst_local[gd];
						it.src_box = src_local[gs];
						const int64_t nc = cells(it);
						biggest = std::max(biggest, nc);
						if (dst_mine && src_mine && loopback) { // sent to and received from this rank itself: the same offset on either side
							PcPeer &pp = peers[my_rank];
							pp.rank = my_rank;
							it.offset = pp.recv_count;
							pp.recv_count += nc * ncomp;
							pp.send_count += nc * ncomp;
							pp.max_recv_cells = std::max(pp.max_recv_cells, nc);
							pp.max_send_cells = std::max(pp.max_send_cells, nc);
							pp.recv.push_back(it);
							pp.send.push_back(it);
						} else if (dst_mine && src_mine) {
							P->local.push_back(it);
							P->max_local_cells = std::max(P->max_local_cells, nc);
						} else if (dst_mine) {
							PcPeer &pp = peers[src_owner[gs]];
							pp.rank = src_owner[gs];
							it.offset = pp.recv_count;
							pp.recv_count += nc * ncomp;
							pp.max_recv_cells = std::max(pp.max_recv_cells, nc);
							pp.recv.push_back(it);
						} else {
							PcPeer &pp = peers[dst_owner[gd]];
							pp.rank = dst_owner[gd];
							it.offset = pp.send_count;
							pp.send_count += nc * ncomp;
							pp.max_send_cells = std::max(pp.max_send_cells, nc);
							pp.send.push_back(it);
						}
					}
				}
			}
		}
	}
	if (biggest * ncomp >= (int64_t{1} << 31)) {
		delete P;
		return setError(ctx, QK_ERR_UNSUPPORTED, "qk_pcopy_plan_create: a region holds 2^31 or more values");
	}
	auto upload = [&](std::vector<PcItem> const &v, PcItem **d) -> int {
		*d = nullptr;
		if (v.empty() || ctx->device < 0) {
			return QK_OK;
		}
		QK_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void **>(d), sizeof(PcItem) * v.size()));
		QK_HIP_CHECK(ctx, hipMemcpy(*d, v.data(), sizeof(PcItem) * v.size(), hipMemcpyHostToDevice));
		return QK_OK;
	};
	P->local_groups = addGroups(P->local);
	int rc = upload(P->local, &P->d_local);
	for (auto &kv : peers) {
		kv.second.recv_groups = addGroups(kv.second.recv); // (the offsets into the peer buffer travel with the items)
		if (rc == QK_OK) {
			rc = upload(kv.second.send, &kv.second.d_send);
		}
		if (rc == QK_OK) {
			rc = upload(kv.second.recv, &kv.second.d_recv);
		}
		P->peers.push_back(kv.second);
	}
	if (rc != QK_OK) {
		qk_pcopy_plan_destroy(P);
		return rc;
	}
	*plan_out = P;
	return QK_OK;
}

int qk_pcopy_plan_destroy(qk_pcopy_plan *plan)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	(void)hipFree(plan->d_local);
	for (auto &p : plan->peers) {
		(void)hipFree(p.d_send);
		(void)hipFree(p.d_recv);
	}
	delete plan;
	return QK_OK;
}

int qk_pcopy_plan_num_peers(qk_pcopy_plan *plan) { return plan == nullptr ? QK_ERR_INVALID : static_cast<int>(plan->peers.size()); }

int qk_pcopy_plan_peer(qk_pcopy_plan *plan, int k, int *rank, int64_t *send_count, int64_t *recv_count)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(plan->ctx, k >= 0 && k < static_cast<int>(plan->peers.size()) && rank && send_count && recv_count, "qk_pcopy_plan_peer: bad index");
	*rank = plan->peers[k].rank;
	*send_count = plan->peers[k].send_count;
	*recv_count = plan->peers[k].recv_count;
	return QK_OK;
}

int qk_pcopy_plan_num_items(qk_pcopy_plan *plan, int kind, int k)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	if (kind == 0) {
		return static_cast<int>(plan->local.size());
	}
	if (k < 0 || k >= static_cast<int>(plan->peers.size()) || (kind != 1 && kind != 2)) {
		return QK_ERR_INVALID;
	}
	return static_cast<int>((kind == 1 ? plan->peers[k].send : plan->peers[k].recv).size());
}

int qk_pcopy_plan_item(qk_pcopy_plan *plan, int kind, int k, int idx, int *dst_box, int *src_box, int lo[3], int hi[3], int shift[3], int64_t *offset)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	const int n = qk_pcopy_plan_num_items(plan, kind, k);
	QK_REQUIRE(plan->ctx, n >= 0 && idx >= 0 && idx < n && dst_box && src_box && lo && hi && shift && offset, "qk_pcopy_plan_item: bad argument");
	PcItem const &it = (kind == 0) ? plan->local[idx] : (kind == 1 ? plan->peers[k].send[idx] : plan->peers[k].recv[idx]);
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

int qk_ParallelCopy_local(qk_pcopy_plan *plan, qk_stream s, const qk_array4 *src_t, qk_array4 *dst_t, int scomp_src, int scomp_dst, int add)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->ctx;
	QK_REQUIRE(ctx, src_t && dst_t && scomp_src >= 0 && scomp_dst >= 0, "ParallelCopy_local: bad argument");
	const int ncomp = plan->ncomp;
	if (plan->local.empty()) {
		return QK_OK;
	}
	ProfScope ps(ctx, static_cast<hipStream_t>(s), "pcopy_local");
	if (add != 0) {
		for (size_t g = 0; g + 1 < plan->local_groups.size(); ++g) {
			forItemChunks(plan->local_groups[g], plan->local_groups[g + 1], [&](int first, int n) {
				hipLaunchKernelGGL((k_pcopy<PC_LOCAL, true>), gridFor(plan->max_local_cells * ncomp, n), dim3(256), 0, static_cast<hipStream_t>(s),
						   plan->d_local + first, src_t, dst_t, static_cast<double *>(nullptr), ncomp, scomp_src, scomp_dst);
			});
		}
	} else {
		forItemChunks(0, static_cast<int>(plan->local.size()), [&](int first, int n) {
			hipLaunchKernelGGL((k_pcopy<PC_LOCAL, false>), gridFor(plan->max_local_cells * ncomp, n), dim3(256), 0, static_cast<hipStream_t>(s),
					   plan->d_local + first, src_t, dst_t, static_cast<double *>(nullptr), ncomp, scomp_src, scomp_dst);
		});
	}
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

int qk_ParallelCopy_pack(qk_pcopy_plan *plan, qk_stream s, int k, const qk_array4 *src_t, int scomp_src, double *sendbuf)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->ctx;
	QK_REQUIRE(ctx, k >= 0 && k < static_cast<int>(plan->peers.size()) && src_t && sendbuf && scomp_src >= 0, "ParallelCopy_pack: bad argument");
	const int ncomp = plan->ncomp;
	PcPeer &pp = plan->peers[k];
	if (pp.send.empty()) {
		return QK_OK;
	}
	forItemChunks(0, static_cast<int>(pp.send.size()), [&](int first, int n) {
		hipLaunchKernelGGL((k_pcopy<PC_PACK, false>), gridFor(pp.max_send_cells * ncomp, n), dim3(256), 0, static_cast<hipStream_t>(s), pp.d_send + first, src_t,
				   static_cast<qk_array4 *>(nullptr), sendbuf, ncomp, scomp_src, 0);
	});
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

int qk_ParallelCopy_unpack(qk_pcopy_plan *plan, qk_stream s, int k, qk_array4 *dst_t, int scomp_dst, const double *recvbuf, int add)
{
	if (plan == nullptr) {
		return QK_ERR_INVALID;
	}
	qk_ctx *ctx = plan->ctx;
	QK_REQUIRE(ctx, k >= 0 && k < static_cast<int>(plan->peers.size()) && dst_t && recvbuf && scomp_dst >= 0, "ParallelCopy_unpack: bad argument");
	const int ncomp = plan->ncomp;
	PcPeer &pp = plan->peers[k];
	if (pp.recv.empty()) {
		return QK_OK;
	}
	if (add != 0) {
		for (size_t g = 0; g + 1 < pp.recv_groups.size(); ++g) {
			forItemChunks(pp.recv_groups[g], pp.recv_groups[g + 1], [&](int first, int n) {
				hipLaunchKernelGGL((k_pcopy<PC_UNPACK, true>), gridFor(pp.max_recv_cells * ncomp, n), dim3(256), 0, static_cast<hipStream_t>(s), pp.d_recv + first,
						   static_cast<const qk_array4 *>(nullptr), dst_t, const_cast<double *>(recvbuf), ncomp, 0, scomp_dst);
			});
		}
	} else {
		forItemChunks(0, static_cast<int>(pp.recv.size()), [&](int first, int n) {
			hipLaunchKernelGGL((k_pcopy<PC_UNPACK, false>), gridFor(pp.max_recv_cells * ncomp, n), dim3(256), 0, static_cast<hipStream_t>(s), pp.d_recv + first,
					   static_cast<const qk_array4 *>(nullptr), dst_t, const_cast<double *>(recvbuf), ncomp, 0, scomp_dst);
		});
	}
	QK_HIP_CHECK(ctx, hipGetLastError());
	return QK_OK;
}

} // extern "C"
