// qk_cooling.hip — the Strang-split optically-thin cooling source from Cloudy tables ("Strang cooling sources").
//   qk_cloudy_tables_read    readCloudyData (reference src/cooling/TabulatedCooling.cpp:9-31, CloudyDataReader.cpp:24-200)
//   qk_cooling_tabulated     computeCooling<problem_t> (src/cooling/TabulatedCooling.hpp:258-317)
//   qk_cooling_evaluate      the per-cell functions a problem calls from its own kernels (ComputeTgasFromEgas, ComputeMMW, ComputeCoolingLength,
//                            cloudy_cooling_function, ComputeEgasFromTgas; TabulatedCooling.hpp:82-220) over arrays, for hosts without device lambdas
// The number of Heun substeps of a cell is data dependent (a few in the hot wind, hundreds at the cloud's cooling front): the kernel runs persistent
// lanes over a queue of cells (k_cooling_tabulated).  The three 25 x 161 tables (97 KB) stay in L2; every lookup is four loads at a computed index.
#include "qk_cooling_device.hpp"
#include "qk_device.hpp"
#include "qk_hdf5_mini.hpp"
#include "qk_internal.hpp"

#include <cstdlib>

using namespace qk;

namespace
{

constexpr double M_H = 1.67262192369e-24 + 9.1093837015e-28; // C::m_p + C::m_e (fundamental_constants.H, CODATA 2018 cgs)

auto tablesOf(const qk_cloudy_tables *t) -> cool::Tables
{
	cool::Tables r{};
	r.log_nH = t->log_nH;
	r.log_T = t->log_Tgas;
	r.cool = t->cooling;
	r.heat = t->heating;
	r.mmw = t->mean_mol_weight;
	r.n_nH = t->n_nH;
	r.n_T = t->n_Tgas;
	r.T_min = t->T_min;
	r.T_max = t->T_max;
	r.mmw_min = t->mmw_min;
	r.mmw_max = t->mmw_max;
	r.m_H = M_H;
	r.k_B = Eos::k_B;
	r.prepared = 0; // (the arrays are in device memory: the kernels call cool::prepare)
	return r;
}

// Persistent lanes over a queue of cells.  The number of Heun substeps of a cell is data dependent (a few in the hot wind, hundreds at a cooling
// front): with one cell per thread a wave runs as long as its slowest lane while the others idle (6x on a log-uniform multiphase medium).  Here a
// lane that finishes its cell stores it and takes the next cell index from a global counter; every trip of the loop is one attempt of a substep
// (cool::heunAttempt) of whatever cell the lane holds, so the lanes of a wave stay busy until the queue is empty.  Cells are numbered box by box
// with the largest box's extents (indices outside a smaller box are skipped).
__global__ void __launch_bounds__(256) k_cooling_tabulated(const qk_box *boxes, int nboxes, int max0, int max1, int max2, qk_array4 *state_t, cool::Tables tab, double gamma,
							   double dt, double T_floor, long long *counters, unsigned long long *queue)
{
	cool::prepare(tab); // (axis ends and spacing: once per lane)
	const long long perBox = static_cast<long long>(max0) * max1 * max2;
	const long long total = perBox * nboxes;
	const double reltol_floor = 0.01, rtol = 1.0e-4;
	cool::HeunState s;
	s.nsteps = 0;
	bool have = false;
	cool::CellCool cc{};
	double Eint0 = 0.0, abstol = 0.0;
	WA4::GT *pE = nullptr, *pEint = nullptr; // (global address space: plain global_load / global_store)
	int mx = 0;
	long long sum = 0;
	bool more = true;
	while (true) {
		if (!have && more) {
			// take the next cell that exists
			while (true) {
				const long long id = static_cast<long long>(atomicAdd(queue, 1ULL));
				if (id >= total) {
					more = false;
					break;
				}
				const int b = static_cast<int>(id / perBox);
				const long long r = id - b * perBox;
				const int k = static_cast<int>(r / (static_cast<long long>(max0) * max1));
				const int r2 = static_cast<int>(r - static_cast<long long>(k) * max0 * max1);
				const int j = r2 / max0;
				const int i = r2 - j * max0;
				const qk_box bx = boxes[b];
				if (i > bx.hi[0] - bx.lo[0] || j > bx.hi[1] - bx.lo[1] || k > bx.hi[2] - bx.lo[2]) {
					continue;
				}
				WA4 S(state_t[b]);
				const int64_t c = S.idx(bx.lo[0] + i, bx.lo[1] + j, bx.lo[2] + k);
				const double rho = S.p[c + S.ns * RHO];
				const double px = S.p[c + S.ns * MX], py = S.p[c + S.ns * MY], pz = S.p[c + S.ns * MZ];
				pE = &S.p[c + S.ns * ENE];
				pEint = &S.p[c + S.ns * EINT];
				// RadSystem::ComputeEintFromEgas (radiation_system.hpp:1288-1297)
				const double Ekin = (px * px + py * py + pz * pz) / (2.0 * rho);
				Eint0 = *pE - Ekin;
				cc = cool::cellCool(tab, rho, gamma);
				abstol = reltol_floor * cool::egasFromTgasAt(tab, rho, cc.X, T_floor, gamma);
				cool::heunBegin(tab, cc, Eint0, dt, s);
				have = true;
				break;
			}
		}
		if (!__any(have)) {
			break; // the queue is empty and every lane of the wave has stored its last cell
		}
		if (have) {
			cool::heunAttempt(tab, cc, dt, rtol, abstol, s);
			if (s.nsteps >= 0) {
				const double dEint = s.E - Eint0;
				*pE += dEint;
				*pEint += dEint;
				mx = max(mx, s.nsteps);
				sum += s.nsteps;
				have = false;
			}
		}
	}
	// max and sum of the substep counts (the reference's iMultiFab max / sum)
	for (int off = 32; off > 0; off >>= 1) {
		mx = max(mx, __shfl_xor(mx, off));
		sum += __shfl_xor(sum, off);
	}
	if ((threadIdx.x & 63) == 0 && sum > 0) {
		atomicMax(&counters[0], static_cast<long long>(mx));
		atomicAdd(reinterpret_cast<unsigned long long *>(&counters[1]), static_cast<unsigned long long>(sum));
	}
}

__global__ void __launch_bounds__(256) k_cooling_evaluate(cool::Tables tab, double gamma, int what, int64_t n, const double *rho, const double *val, double *out)
{
	const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (t >= n) {
		return;
	}
	cool::prepare(tab);
	switch (what) {
	case QK_COOLING_TGAS_FROM_EGAS:
		out[t] = cool::tgasFromEgas(tab, rho[t], val[t], gamma);
		break;
	case QK_COOLING_EGAS_FROM_TGAS:
		out[t] = cool::egasFromTgas(tab, rho[t], val[t], gamma);
		break;
	case QK_COOLING_MMW:
		out[t] = cool::meanMolecularWeight(tab, rho[t], val[t], gamma);
		break;
	case QK_COOLING_LENGTH:
		out[t] = cool::coolingLength(tab, rho[t], val[t], gamma);
		break;
	default:
		out[t] = cool::netHeating(tab, rho[t], val[t]);
		break;
	}
}

template <class T> auto hostCopy(std::vector<T> const &v) -> T *
{
	auto *p = static_cast<T *>(std::malloc(sizeof(T) * v.size()));
	if (p != nullptr) {
		std::copy(v.begin(), v.end(), p);
	}
	return p;
}

} // namespace

int qk_cloudy_tables_read(qk_ctx *ctx, const char *path, qk_cloudy_tables *out)
{
	QK_REQUIRE(ctx, path != nullptr && out != nullptr, "qk_cloudy_tables_read: NULL argument");
	*out = qk_cloudy_tables{};
	try {
		h5::File const file(path);
		h5::Dataset const coolds = file.dataset("Cooling");
		// grid rank and extents from the attributes of /Cooling (CloudyDataReader.cpp:68-89)
		auto const rankAttr = coolds.attrs.find("Rank");
		auto const dimAttr = coolds.attrs.find("Dimension");
		if (rankAttr == coolds.attrs.end() || dimAttr == coolds.attrs.end()) {
			return setError(ctx, QK_ERR_INVALID, "qk_cloudy_tables_read", "/Cooling carries no Rank / Dimension attributes");
		}
		int64_t const rank = h5::asIntegers(rankAttr->second.type, rankAttr->second.raw, 1)[0];
		if (rank != 2) {
			return setError(ctx, QK_ERR_UNSUPPORTED, "qk_cloudy_tables_read", "the table must have rank 2 (density, temperature): CLOUDY_MAX_DIMENSION");
		}
		auto const dims = h5::asIntegers(dimAttr->second.type, dimAttr->second.raw, 2);
		int const n_nH = static_cast<int>(dims[0]), n_T = static_cast<int>(dims[1]);
		h5::Dataset const p1 = file.dataset("Parameter1"), temp = file.dataset("Temperature");
		std::vector<double> log_nH = h5::asDoubles(p1.type, p1.raw, static_cast<size_t>(n_nH));
		std::vector<double> log_T = h5::asDoubles(temp.type, temp.raw, static_cast<size_t>(n_T));
		// the temperature axis is stored in K: its range is kept, the axis becomes log10 T (:103-110)
		double T_min = std::numeric_limits<double>::max(), T_max = std::numeric_limits<double>::min();
		for (auto &T : log_T) {
			T_min = std::min(T, T_min);
			T_max = std::max(T, T_max);
			T = std::log10(T);
		}
		// cooling and heating rates: FastMath::log10 of the rate in units of CoolUnit (cgs code units: x^2 m_h^2 / (t^3 d) with m_h = 1.67e-24),
		// 1e-99 / CoolUnit where the table holds no positive value (:42-50,137-141)
		double const mh = 1.67e-24;
		double const CoolUnit = (1.0 * 1.0 * mh * mh) / (1.0 * 1.0 * 1.0 * 1.0);
		double const small = cool::fastLog10(1.0e-99 / CoolUnit);
		size_t const ntab = static_cast<size_t>(n_nH) * static_cast<size_t>(n_T);
		auto rates = [&](char const *name) {
			h5::Dataset const ds = file.dataset(name);
			if (ds.count() != ntab) {
				throw std::runtime_error(std::string("hdf5: /") + name + " does not have Dimension[0] x Dimension[1] values");
			}
			return h5::asDoubles(ds.type, ds.raw, ntab);
		};
		// the file stores [density][temperature] (temperature fastest); the kernels index [temperature][density] (density fastest), the layout
		// of the reference's extract_2d_table (:206-222)
		auto transposed = [&](std::vector<double> const &file_order) {
			std::vector<double> t(ntab);
			for (int i = 0; i < n_nH; ++i) {
				for (int j = 0; j < n_T; ++j) {
					t[static_cast<size_t>(i) + static_cast<size_t>(n_nH) * static_cast<size_t>(j)] =
					    file_order[static_cast<size_t>(j) + static_cast<size_t>(n_T) * static_cast<size_t>(i)];
				}
			}
			return t;
		};
		std::vector<double> coolv = rates("Cooling"), heatv = rates("Heating"), mmwv = rates("MMW");
		for (auto *v : {&coolv, &heatv}) {
			for (auto &x : *v) {
				double const value = x / CoolUnit;
				x = value > 0 ? cool::fastLog10(value) : small;
			}
		}
		double mmw_min = std::numeric_limits<double>::max(), mmw_max = std::numeric_limits<double>::min();
		for (double const m : mmwv) {
			mmw_min = std::min(m, mmw_min);
			mmw_max = std::max(m, mmw_max);
		}
		out->n_nH = n_nH;
		out->n_Tgas = n_T;
		out->log_nH = hostCopy(log_nH);
		out->log_Tgas = hostCopy(log_T);
		out->cooling = hostCopy(transposed(coolv));
		out->heating = hostCopy(transposed(heatv));
		out->mean_mol_weight = hostCopy(transposed(mmwv));
		out->T_min = T_min;
		out->T_max = T_max;
		out->mmw_min = mmw_min;
		out->mmw_max = mmw_max;
	} catch (std::exception const &e) {
		return setError(ctx, QK_ERR_INVALID, "qk_cloudy_tables_read", e.what());
	}
	return QK_OK;
}

int qk_cloudy_tables_free(qk_cloudy_tables *t)
{
	if (t == nullptr) {
		return QK_ERR_INVALID;
	}
	for (const double *p : {t->log_nH, t->log_Tgas, t->cooling, t->heating, t->mean_mol_weight}) {
		std::free(const_cast<double *>(p));
	}
	*t = qk_cloudy_tables{};
	return QK_OK;
}

int qk_cooling_tabulated(qk_level *lev, qk_stream s, const qk_hydro_traits *t, qk_array4 *state_t, const qk_cloudy_tables *device_tables, double dt, double T_floor,
			 long long *d_counters)
{
	if (lev == nullptr) {
		return QK_ERR_INVALID;
	}
	QK_REQUIRE(lev->ctx, t != nullptr && state_t != nullptr && device_tables != nullptr && d_counters != nullptr, "cooling_tabulated: NULL argument");
	QK_REQUIRE(lev->ctx, device_tables->n_nH >= 2 && device_tables->n_Tgas >= 2, "cooling_tabulated: the tables need at least two points per axis");
	if (lev->nboxes == 0) {
		return QK_OK;
	}
	// the queue word of the persistent kernel: owned by the LEVEL, cleared on the stream before every launch (the calls of one level are ordered by
	// its stream; levels on different streams do not share a counter)
	if (lev->d_cooling_queue == nullptr) {
		void *p = nullptr;
		if (hipMalloc(&p, sizeof(unsigned long long)) != hipSuccess) {
			return setError(lev->ctx, QK_ERR_HIP, "cooling_tabulated", "cannot allocate the queue word");
		}
		lev->d_cooling_queue = static_cast<unsigned long long *>(p);
	}
	unsigned long long *queue = lev->d_cooling_queue;
	QK_HIP_CHECK(lev->ctx, hipMemsetAsync(queue, 0, sizeof(unsigned long long), static_cast<hipStream_t>(s)));
	// enough resident waves to fill the chip (the kernel holds ~100 VGPRs: 4 waves per SIMD), never more lanes than cells
	const long long cells = static_cast<long long>(lev->maxlen[0]) * lev->maxlen[1] * lev->maxlen[2] * lev->nboxes;
	const unsigned blocks = static_cast<unsigned>(std::max<long long>(1, std::min<long long>(256LL * 8, (cells + 255) / 256)));
	ProfScope ps(lev->ctx, static_cast<hipStream_t>(s), "cooling_tabulated");
	hipLaunchKernelGGL(k_cooling_tabulated, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(s), lev->d_boxes, lev->nboxes, lev->maxlen[0], lev->maxlen[1],
			   lev->maxlen[2], state_t, tablesOf(device_tables), t->gamma, dt, T_floor, d_counters, queue);
	const hipError_t e = hipGetLastError();
	return (e == hipSuccess) ? QK_OK : setError(lev->ctx, QK_ERR_HIP, "cooling_tabulated", hipGetErrorString(e));
}

int qk_cooling_evaluate(qk_ctx *ctx, qk_stream s, const qk_cloudy_tables *device_tables, double gamma, int what, int64_t n, const double *d_rho, const double *d_val,
			double *d_out)
{
	QK_REQUIRE(ctx, device_tables != nullptr && d_rho != nullptr && d_val != nullptr && d_out != nullptr, "cooling_evaluate: NULL argument");
	QK_REQUIRE(ctx, what >= QK_COOLING_TGAS_FROM_EGAS && what <= QK_COOLING_NET_HEATING, "cooling_evaluate: unknown quantity");
	if (n <= 0) {
		return QK_OK;
	}
	hipLaunchKernelGGL(k_cooling_evaluate, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(s), tablesOf(device_tables), gamma, what,
			   n, d_rho, d_val, d_out);
	const hipError_t e = hipGetLastError();
	return (e == hipSuccess) ? QK_OK : setError(ctx, QK_ERR_HIP, "cooling_evaluate", hipGetErrorString(e));
}
