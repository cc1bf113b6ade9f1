/* oracle/ref_nanoflann.cpp -- TEST INFRASTRUCTURE.
 * Thin driver around the REFERENCE's own vendored radius search
 * (/root/reference/thirdparty/cvo/thirdparty/nanoflann.hpp and
 * KDTreeVectorOfVectorsAdaptor.h), compiled from where those headers lie into
 * oracle/_ref/libref_nanoflann.so (git-ignored).  It calls them exactly the way
 * cvo.cpp:133-148 does (float, metric_L2, leaf size 10, default SearchParams
 * => results sorted by distance) so the oracle's own exact radius search can be
 * pinned against real reference code.  No reference source is copied.
 * The reference's cloud_t is std::vector<Eigen::Vector3f> (data_type.h:30);
 * Eigen is absent here, and the adaptor is a template over any vector-of-vectors
 * type, so std::array<float,3> is used as the element type. */
#include <array>
#include <cstddef>
#include <utility>
#include <vector>
#include <nanoflann.hpp>
#include <KDTreeVectorOfVectorsAdaptor.h>

typedef std::vector<std::array<float, 3>> cloud_t;
typedef KDTreeVectorOfVectorsAdaptor<cloud_t, float> kd_tree_t;

extern "C" int ref_radius_search(const float* cloud_xyz, int n, const float* queries, int nq, float radius_sq,
                                 int* out_count, int* out_idx, float* out_d2, int cap_per_query) {
    cloud_t cloud(n);
    for (int i = 0; i < n; ++i) cloud[i] = {cloud_xyz[i * 3], cloud_xyz[i * 3 + 1], cloud_xyz[i * 3 + 2]};
    kd_tree_t mat_index(3 /*dim*/, cloud, 10 /* max leaf */);
    mat_index.index->buildIndex();
    for (int q = 0; q < nq; ++q) {
        const float search_radius = radius_sq;
        std::vector<std::pair<size_t, float>> ret_matches;
        nanoflann::SearchParams params;
        const size_t nMatches = mat_index.index->radiusSearch(queries + q * 3, search_radius, ret_matches, params);
        out_count[q] = (int)nMatches;
        for (size_t k = 0; k < nMatches && (int)k < cap_per_query; ++k) {
            out_idx[(size_t)q * cap_per_query + k] = (int)ret_matches[k].first;
            out_d2[(size_t)q * cap_per_query + k] = ret_matches[k].second;
        }
    }
    return 0;
}
