// integration/hip_toed.cpp -- reference-side binding, replaces src/toed/cpu_toed.cpp in the reference's
// src/CMakeLists.txt.  The header include/toed/cpu_toed.hpp stays untouched, so Pipeline.h:96,193 and
// Pipeline::ProcessEdges (src/Pipeline.cpp:24-29) compile and behave as before.
//
// Not BUILT in this repository (needs the reference tree + OpenCV).  Where the reference tree is present it is checked with
// g++ -fsyntax-only against the reference's real headers (tests/test_integration_syntax.py: signatures and member names
// agree); the same marshalling is compiled and parity-tested here through include/ebvo/adapters.hpp
// (tests/test_cpp_adapter.py, tests/test_cpp_stagewise.py).
#include <mutex>
#include <unordered_map>

#include "../include/definitions.h"
#include "../include/toed/cpu_toed.hpp"
#include "ebvo/adapters.hpp" // -I<this repo>/include

namespace
{
// The class has no spare member for a device handle and its header is kept byte-identical, so the
// handle lives in a side table keyed by the object.
std::mutex g_mu;
std::unordered_map<const ThirdOrderEdgeDetectionCPU *, std::unique_ptr<ebvo::ThirdOrderEdgeDetectionHIP<Edge>>> g_impl;
ebvo::ThirdOrderEdgeDetectionHIP<Edge> &impl(const ThirdOrderEdgeDetectionCPU *self)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return *g_impl.at(self);
}
} // namespace

// the stereo / temporal matchers borrow the detector's device context (integration/stereo_matches_hip.cpp)
ebvo::Context::Ptr ebvo_context_of(const ThirdOrderEdgeDetectionCPU *toed) { return impl(toed).context(); }

ThirdOrderEdgeDetectionCPU::ThirdOrderEdgeDetectionCPU(int H, int W)
{
    img_height = H;
    img_width = W;
    kernel_sz = TOED_KERNEL_SIZE;
    shifted_kernel_sz = kernel_sz + 2;
    g_sig = TOED_SIGMA;
    interp_img_height = H * 2;
    interp_img_width = W * 2;
    omp_threads = 1; // no host threads are used
    img = Ix = Iy = I_grad_mag = I_orient = nullptr; // the maps live in HBM
    subpix_pos_x_map = subpix_pos_y_map = subpix_grad_mag_map = nullptr;
    num_of_edge_data = 4;
    auto p = std::make_unique<ebvo::ThirdOrderEdgeDetectionHIP<Edge>>(H, W);
    subpix_edge_pts_final = p->subpix_edge_pts_final; // N x 4 host array owned by the adapter
    std::lock_guard<std::mutex> lk(g_mu);
    g_impl[this] = std::move(p);
}

ThirdOrderEdgeDetectionCPU::~ThirdOrderEdgeDetectionCPU()
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_impl.erase(this);
}

void ThirdOrderEdgeDetectionCPU::get_Third_Order_Edges(cv::Mat img)
{
    auto &d = impl(this);
    d.get_Third_Order_Edges(img); // cv::Mat has .data/.rows/.cols/.step
    toed_edges = d.toed_edges;    // Edge{location, orientation, index}; b_isEmpty = true, frame_source = -1
    subpix_edge_pts_final = d.subpix_edge_pts_final; // N x 4 rows in page-locked memory of the adapter's context
    Total_Num_Of_TOED = d.Total_Num_Of_TOED;
    edge_pt_list_idx = d.edge_pt_list_idx;
    time_conv = d.time_conv;
    time_nms = d.time_nms;
}

// The three stages are one device pipeline; the individual entry points remain for API compatibility.
void ThirdOrderEdgeDetectionCPU::preprocessing(cv::Mat) { toed_edges.clear(); }
void ThirdOrderEdgeDetectionCPU::convolve_img() {}
int ThirdOrderEdgeDetectionCPU::non_maximum_suppresion() { return Total_Num_Of_TOED; }
void ThirdOrderEdgeDetectionCPU::read_array_from_file(std::string, double *, int, int) {}
void ThirdOrderEdgeDetectionCPU::write_array_to_file(std::string, double *, int, int) {}
