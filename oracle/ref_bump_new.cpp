// ref_bump_new.cpp — TEST INFRASTRUCTURE, linked into the PROGRAM oracle/_ref/orb_ref_cli only (a replacement operator new applies to
// every module of a process only when the program defines it).  ORBextractor::DistributeOctTree sorts pair<int, ExtractorNode*> (ORBextractor.cpp:852): nodes holding equally many
// keypoints are ordered by their HEAP ADDRESS, so the reference's own output depends on the allocator's history (observed
// here: with glibc malloc 3 keypoints on each of two levels change between otherwise identical runs).  The oracle and the product define
// the tie as "node creation order".  A monotonic operator new makes addresses increase in creation order inside the reference binary, so
// the reference's pointer order BECOMES creation order and the two can be compared keypoint for keypoint.  Memory is never returned
// (a test process extracts a handful of frames).
#include <cstdlib>
#include <new>

namespace {
char* g_cur = nullptr;
char* g_end = nullptr;
void* bump(std::size_t n) {
  n = (n + 15) & ~(std::size_t)15;
  if (g_cur == nullptr || (std::size_t)(g_end - g_cur) < n) {
    const std::size_t chunk = n > ((std::size_t)64 << 20) ? n : ((std::size_t)64 << 20);
    g_cur = (char*)std::malloc(chunk);   // chunks themselves come from ascending addresses often, but ties inside one octree call
    if (!g_cur) throw std::bad_alloc();  // never straddle two chunks in practice (a level's nodes take a few hundred KB)
    g_end = g_cur + chunk;
  }
  void* p = g_cur;
  g_cur += n;
  return p;
}
}  // namespace

void* operator new(std::size_t n) { return bump(n); }
void* operator new[](std::size_t n) { return bump(n); }
void operator delete(void*) noexcept {}
void operator delete[](void*) noexcept {}
void operator delete(void*, std::size_t) noexcept {}
void operator delete[](void*, std::size_t) noexcept {}
