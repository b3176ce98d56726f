// demo_vaqhip.cpp -- the query half of the reference's examples/demo_vaq.cpp
// (:58-92 load, :336-363 search + recall) on the MI355X path.  Training is out
// of scope (it needs glpk / armadillo in the reference), so the index is read
// from the files `demo_vaq --save <centroids> --save-enc <codebook>` writes,
// plus the rotation (which the reference never persists) as a raw D x D
// float32 file.
//
//   demo_vaqhip --centroids c.bin --codebook cb.bin [--eigen e.f32] \
//               --queries q.fvecs --timeseries-size 128 [--queries-size N] \
//               --method VAQ64m8min8max8var1,HEAP --k 100 \
//               [--groundtruth gt.ivecs] [--result out.csv] [--bits 8,8,...]
//               [--visit-cluster 0.25]     (demo_vaq.cpp:43,57; with a ...,EA_TI<T>m<seg> method)
//               [--ti-clusters c.f32]      (raw T x seg*L float32; default: random decoded rows)
//               [--refine 100,200 --dataset base.fvecs [--dataset-size N]]   (or --dataset-refine)
//                                          (demo_vaq.cpp:40, :312-345 and scripts/run_demos.sh:9,22: per value R,
//                                           search R >= k candidates, then VAQ::refine re-ranks them against the
//                                           raw vectors; results go to <result>_R<R> when several R are given)
//               [--devices 0,1,2,3]        (shard the rows over these GPUs: RCCL all-gather + merge)
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>

#include "vaqhip_io.hpp"

using namespace vaqhip;

int main(int argc, char **argv) {
  std::map<std::string, std::string> a = {{"k", "100"}, {"method", "VAQ64m8min8max8var1,HEAP"},
                                          {"timeseries-size", "128"}, {"queries-size", "-1"}};
  for (int i = 1; i + 1 < argc; i += 2) {
    if (std::strncmp(argv[i], "--", 2) != 0) { std::cerr << "bad argument " << argv[i] << "\n"; return 2; }
    a[argv[i] + 2] = argv[i + 1];
  }
  for (const char *req : {"centroids", "codebook", "queries"})
    if (!a.count(req)) { std::cerr << "missing --" << req << "\n"; return 2; }
  try {
    VaqHip vaq;
    vaq.parseMethodString(a["method"]);
    vaq.mCentroidsPerSubs = loadCentroids(a["centroids"]);
    vaq.mCodebook = loadCodebook(a["codebook"]);
    const int M = (int)vaq.mCentroidsPerSubs.size();
    if (a.count("bits")) {  // --hc-bitalloc style list (demo_vaq.cpp:94-97)
      std::stringstream ss(a["bits"]);
      std::string t;
      while (std::getline(ss, t, ',')) vaq.mBitsAlloc.push_back(std::atoi(t.c_str()));
    } else {
      for (int s = 0; s < M; s++) vaq.mBitsAlloc.push_back((int)std::lround(std::log2((double)vaq.mCentroidsPerSubs[s].rows())));
    }
    const int D = vaq.mTotalDim();
    const int N = std::atoi(a["timeseries-size"].c_str());
    if (a.count("eigen")) {
      vaq.mEigenVectors = RowMatrixF(D, D);
      detail::File f(a["eigen"], "rb");
      f.read(vaq.mEigenVectors.data(), sizeof(float), (size_t)D * D);
    }
    if (vaq.searchMethod() & VaqHip::NNMethod::TI) {  // demo_vaq.cpp:57, :263-267
      if (a.count("visit-cluster")) vaq.mVisit = (float)std::atof(a["visit-cluster"].c_str());
      if (vaq.mTISegmentNum == -1) vaq.mTISegmentNum = M;
      if (a.count("ti-clusters")) {
        vaq.mTIClusters = RowMatrixF((size_t)vaq.mTIClusterNum, (size_t)vaq.mTISegmentNum * vaq.mSubsLen());
        detail::File f(a["ti-clusters"], "rb");
        f.read(vaq.mTIClusters.data(), sizeof(float), vaq.mTIClusters.rows() * vaq.mTIClusters.cols());
      }
      vaq.clusterTI(false, true);
    }
    RowMatrixF queries = readFVecs(a["queries"], N, std::atoi(a["queries-size"].c_str()), D - N);
    const int k = std::atoi(a["k"].c_str());
    std::cout << "index: " << vaq.mCodebook.rows() << " rows x " << M << " subspaces, D=" << D
              << ", queries " << queries.rows() << ", k=" << k << std::endl;
    if (a.count("devices")) {
      std::vector<int> devs;
      std::stringstream ss(a["devices"]);
      std::string t;
      while (std::getline(ss, t, ',')) devs.push_back(std::atoi(t.c_str()));
      vaq.setDevices(devs);
      std::cout << "sharding the rows over " << devs.size() << " device entr" << (devs.size() == 1 ? "y" : "ies") << std::endl;
    }
    // --refine R1,R2,... (demo_vaq.cpp:312-323); without it one plain search (refine = 0)
    std::vector<int> refines;
    if (a.count("refine")) {
      std::stringstream ss(a["refine"]);
      std::string t;
      while (std::getline(ss, t, ',')) refines.push_back(std::atoi(t.c_str()));
    }
    if (refines.empty()) refines.push_back(0);
    RowMatrixF datasetrefine;
    bool any_refine = false;
    for (const int r : refines) any_refine = any_refine || r >= k;
    if (any_refine) {
      // the reference re-reads --dataset for this (demo_vaq.cpp:320-333); --dataset-refine names another file
      const std::string raw = a.count("dataset-refine") ? a["dataset-refine"] : (a.count("dataset") ? a["dataset"] : "");
      if (raw.empty()) throw Error(VAQHIP_EINVAL, "--refine needs the raw vectors: --dataset <.fvecs> (or --dataset-refine)");
      datasetrefine = readFVecs(raw, N, a.count("dataset-size") ? std::atoi(a["dataset-size"].c_str()) : -1, 0);
    }
    RowMatrix<int> gt;
    if (a.count("groundtruth")) gt = readIVecs(a["groundtruth"], k);
    vaq.sync();
    for (const int refine : refines) {
      auto t0 = std::chrono::steady_clock::now();
      const int searchK = refine >= k ? refine : k;  // demo_vaq.cpp:338
      LabelDistVecF answers = vaq.search(queries, searchK, true);
      if (refine >= k) {
        std::cout << "Refining the answer with Refine = " << refine << std::endl;
        // the raw queries (first N dims), as the reference passes them (demo_vaq.cpp:342)
        RowMatrixF qraw((size_t)queries.rows(), (size_t)N);
        for (size_t i = 0; i < qraw.rows(); i++)
          for (size_t j = 0; j < qraw.cols(); j++) qraw(i, j) = queries(i, j);
        answers = vaq.refine(qraw, answers, datasetrefine, k);
      }
      double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      std::cout << "== Querying time: " << sec << " s (" << queries.rows() / sec
                << " queries/s, host buffers in and out)" << std::endl;
      if (a.count("result")) {
        std::string fp = a["result"];
        if (refines.size() > 1) fp += "_R" + std::to_string(refine);  // demo_vaq.cpp:349-351
        writeKNNResults(fp, answers, queries.rows());
      }
      if (a.count("groundtruth"))
        std::cout << "\tprecision(avg_recall): " << getAvgRecall(answers.labels, gt, k)
                  << "\n\trecall@R: " << getRecallAtR(answers.labels, gt, k) << std::endl;
    }
  } catch (const std::exception &e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
