// TEST INFRASTRUCTURE — pins oracle.bow_transform (oracle/match_ref.cpp: ora_bow_transform) to the reference's vendored DBoW2:
// cslam/thirdparty/DBoW2/DBoW2/{TemplatedVocabulary.h, FORB.cpp, ScoringObject.cpp, BowVector.cpp, FeatureVector.cpp} and DUtils/{Random,Timestamp}.cpp are compiled
// verbatim (oracle/Makefile.ref); this file only does what KeyFrame::ComputeBoW / Frame::ComputeBoW do (KeyFrame.cpp:277-286): Converter::toDescriptorVector's
// row split, then ORBVocabulary::transform(descs, BowVector, FeatureVector, levelsup).  The vocabulary is loaded with the reference's own loadFromTextFile
// (the ORBvoc.txt format: "k L scoring weighting", then one "parent isLeaf 32-bytes weight" line per node).
#include <cslam/ORBVocabulary.h>

#include <cstdint>
#include <cstring>

extern "C" {

void* ref_vocab_load_text(const char* path) {
  cslam::ORBVocabulary* v = new cslam::ORBVocabulary();
  if (!v->loadFromTextFile(path) || v->empty()) { delete v; return nullptr; }
  return v;
}
void ref_vocab_free(void* h) { delete static_cast<cslam::ORBVocabulary*>(h); }
int ref_vocab_size(void* h) { return (int)static_cast<cslam::ORBVocabulary*>(h)->size(); }

// Returns the BowVector size; bow_ids / bow_vals in the map's (ascending word) order.  FeatureVector as CSR: fv_nodes[n_fv], fv_off[n_fv + 1], fv_feat[...]
// (n_fv written to *n_fv_out).  Buffers sized N by the caller.
int ref_bow_transform(void* h, const uint8_t* desc, int N, int levelsup, int32_t* bow_ids, double* bow_vals, int32_t* n_fv_out, int32_t* fv_nodes, int32_t* fv_off,
                      int32_t* fv_feat) {
  const cslam::ORBVocabulary* voc = static_cast<cslam::ORBVocabulary*>(h);
  cv::Mat D(N, 32, CV_8U);
  if (N) std::memcpy(D.data, desc, (size_t)N * 32);
  std::vector<cv::Mat> vDesc;                      // Converter::toDescriptorVector (Converter.cc:41-49)
  vDesc.reserve(D.rows);
  for (int j = 0; j < D.rows; j++) vDesc.push_back(D.row(j));
  DBoW2::BowVector bv; DBoW2::FeatureVector fv;
  voc->transform(vDesc, bv, fv, levelsup);
  int n = 0;
  for (DBoW2::BowVector::const_iterator it = bv.begin(); it != bv.end(); ++it, ++n) { bow_ids[n] = (int32_t)it->first; bow_vals[n] = it->second; }
  int k = 0, w = 0;
  fv_off[0] = 0;
  for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++k) {
    fv_nodes[k] = (int32_t)it->first;
    for (unsigned int f : it->second) fv_feat[w++] = (int32_t)f;
    fv_off[k + 1] = w;
  }
  *n_fv_out = k;
  return n;
}

// L1 score between two BowVectors as the reference's scoring object computes it (ScoringObject.cpp:25-55), for the KeyFrameDatabase-side consumers
double ref_bow_score(void* h, const int32_t* ids1, const double* vals1, int n1, const int32_t* ids2, const double* vals2, int n2) {
  const cslam::ORBVocabulary* voc = static_cast<cslam::ORBVocabulary*>(h);
  DBoW2::BowVector a, b;
  for (int i = 0; i < n1; i++) a.insert(a.end(), std::make_pair((DBoW2::WordId)ids1[i], vals1[i]));
  for (int i = 0; i < n2; i++) b.insert(b.end(), std::make_pair((DBoW2::WordId)ids2[i], vals2[i]));
  return voc->score(a, b);
}

}  // extern "C"
