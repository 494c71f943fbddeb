// ccm_host_c.h — the C entry points of libccm_host.so (the host mirror of the reference's class API, ccm_host.h / ccm_host.cpp): flat arrays in, flat arrays out.
// Used by the drop-in translation units under shim/ (which include the REFERENCE's class headers and therefore cannot include ccm_host.h, whose classes carry
// the same names) and by the Python harness.  Every entry returns -1000 when the device path throws (missing library, HIP error): there is no CPU fallback.
// `device` selects the GPU; each calling thread keeps one context per device for its lifetime (SURVEY 8b threading).
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int ccmh_search_by_projection_mp(int device, const float* kx, const float* ky, const int32_t* oct, const uint8_t* fdesc, int N, float minX, float minY, float maxX, float maxY, const float* scale_factors, int n_mp, const uint8_t* in_view, const float* px, const float* py, const int32_t* lvl, const float* vcos, const uint8_t* mp_desc, float th, float nnratio, int32_t* frame_mp);
int ccmh_search_by_projection_last(int device, const float* kx, const float* ky, const int32_t* oct, const float* kangle, const uint8_t* fdesc, int N, float minX, float minY, float maxX, float maxY, const float* scale_factors, int n_last, const uint8_t* valid, const float* u, const float* v, const int32_t* l_oct, const float* l_angle, const uint8_t* l_desc, float th, int check_ori, int32_t* cur_mp);
int ccmh_local_ba(int device, int n_cam, int n_pt, int n_edge, double* cam_qt, const uint8_t* cam_fixed, const double* cam_K, double* pt_xyz, const int32_t* e_cam, const int32_t* e_pt, const double* e_obs, const double* e_info, uint8_t* to_erase);
int ccmh_search_by_projection_mp_dev(int device, const float* K, const float* dist, int n_dist, int w, int h, const void* kps_raw, const uint8_t* fdesc, int N, const float* scale_factors, int n_mp, const uint8_t* in_view, const float* px, const float* py, const int32_t* lvl, const float* vcos, const uint8_t* mp_desc, float th, float nnratio, int32_t* frame_mp, float* xy_un_out);
int ccmh_search_by_projection_last_dev(int device, const void* kps_un, const uint8_t* cdesc, int N, int w, int h, const float* scale_factors, int n_last, const uint8_t* valid, const float* u, const float* v, const int32_t* oct, const float* angle, const uint8_t* mp_desc, float th, int check_ori, int32_t* frame_mp);
int ccmh_optimize_sim3(int device, double* sim3, int n, const double* P1c, const double* P2c, const double* obs1, const double* obs2, const double* info1, const double* info2, const double* K1, const double* K2, float th2, int fix_scale, uint8_t* keep);
int ccmh_orb_extract(int device, int nfeatures, const uint8_t* img, int w, int h, void* kps_out, uint8_t* desc_out, int cap);
int ccmh_search_bow(int device, int mode, const int32_t* n1, const int32_t* o1, const int32_t* i1, int nn1, const int32_t* n2, const int32_t* o2, const int32_t* i2, int nn2, const uint8_t* has1, const uint8_t* has2, const uint8_t* d1, const float* x1, const float* y1, const float* a1, int N1, const uint8_t* d2, const float* x2, const float* y2, const int32_t* oct2, const float* a2, int N2, const float* F12, float ex, float ey, const float* sigma2_2, const float* sf2, float nnratio, int check_ori, int32_t* out);
/* the SearchForTriangulation fan-out of a new keyframe (Mapping.cpp:335) as ONE device launch: create computes the Hamming tables of all neighbours, resolve replays the
 * reference's sequential rules of the call against neighbour j with the map-point flags as they are at that call (NULL: as at create) */
void* ccmh_tri_batch_create(int device, float nnratio, int check_ori, const int32_t* n1, const int32_t* o1, const int32_t* i1, int nn1, const uint8_t* has1, const uint8_t* d1, const float* x1, const float* y1, const float* a1, int N1, int n_nb, const int32_t* const* n2, const int32_t* const* o2, const int32_t* const* i2, const int32_t* nn2, const uint8_t* const* has2, const uint8_t* const* d2, const float* const* x2, const float* const* y2, const int32_t* const* oct2, const float* const* a2, const int32_t* N2);
int ccmh_tri_batch_resolve(void* h, int j, const uint8_t* has1_now, const uint8_t* has2_now, const float* F12, float ex, float ey, const float* sigma2_2, const float* sf2, int32_t* matches12);
long long ccmh_tri_batch_candidates(void* h);
void ccmh_tri_batch_destroy(void* h);
int ccmh_search_for_initialization(int device, const float* x1, const float* y1, const int32_t* oct1, const float* a1, const uint8_t* d1, int N1, const float* x2, const float* y2, const int32_t* oct2, const float* a2, const uint8_t* d2, int N2, float minX, float minY, float maxX, float maxY, float* prev_xy, int window, float nnratio, int check_ori, int32_t* matches12);
int ccmh_projected_window_search(int device, const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc, int N, float minX, float minY, float maxX, float maxY, const float* scale_factors, const float* inv_sigma2, int n_pts, const uint8_t* valid, const float* u, const float* v, const int32_t* level, const uint8_t* pdesc, float th, int chi2_gate, int dist_threshold, int32_t* matched, int claim, const uint8_t* no_claim, int32_t* best_idx, int32_t* best_dist);
void* ccmh_fuse_batch_create_cand(int device, int S, const int32_t* kf_off, const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc, const float* const* inv_sigma2, const int32_t* pt_off, const uint8_t* valid, const float* u, const float* v, const int32_t* level, const uint8_t* pdesc, const int32_t* cand_off, const int32_t* cand_base, const int32_t* cand_idx, int chi2_gate, int dist_threshold);
int ccmh_fuse_batch_resolve(void* h, int s, const uint8_t* skip_now, int n_pts, int32_t* best_idx, int32_t* best_dist);
void ccmh_fuse_batch_destroy(void* h);
int ccmh_projected_window_search_cand(int device, const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc, int N, const float* inv_sigma2, int n_pts, const uint8_t* valid, const float* u, const float* v, const int32_t* level, const uint8_t* pdesc, const int32_t* cand_off, const int32_t* cand_idx, int chi2_gate, int dist_threshold, int32_t* matched, int claim, const uint8_t* no_claim, int32_t* best_idx, int32_t* best_dist);
int ccmh_projected_window_search_dev(int device, const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc, int N, float maxX, float maxY, const float* scale_factors, const float* inv_sigma2, int n_pts, const uint8_t* valid, const float* u, const float* v, const int32_t* level, const uint8_t* pdesc, float th, int chi2_gate, int dist_threshold, int32_t* matched, int claim, const uint8_t* no_claim, int32_t* best_idx, int32_t* best_dist);
int ccmh_bow_transform(int device, int n_nodes, int L, const int32_t* child_off, const int32_t* child_id, const uint8_t* node_desc, const int32_t* word_id, const double* weight, const uint8_t* desc, int N, int levelsup, int32_t* bow_ids, double* bow_vals, int32_t* fv_nodes, int32_t* fv_off, int32_t* fv_idx, int32_t* sizes );
void ccmh_to_se3quat(const float* Tcw16, double* qt7);
void ccmh_se3quat_to_cvmat(const double* qt7, float* Tcw16);
void ccmh_sim3_to_cvse3(const double* s8, float* Tcw16);
#ifdef __cplusplus
}
#endif
