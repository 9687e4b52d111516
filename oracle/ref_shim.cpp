// ref_shim.cpp -- extern "C" doorway into the REFERENCE's own compiled sources.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; it contains no reference code.  It is
// compiled together with the reference's source files where they lie under /root/reference
// (see oracle/Makefile) into oracle/_ref/libref_{int16,fp32}.so, which stay out of git.
// It lets the tests call the reference's YOLO2_FPGA (hls/models/yolov2/yolo2_accel.hpp:10-17)
// and yolov2_hls_ps (:23) through ctypes to validate oracle/yolo2_oracle.c and to generate
// the fixtures under tests/golden/.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>

#include <core/yolo.h>
#include <core/precision.hpp>
#include <api.hpp>

extern "C" {

int ref_is_int16(void)
{
#ifdef INT16_MODE
    return 1;
#else
    return 0;
#endif
}

// Thin pass-through: same 27 arguments, bool as int.
void ref_YOLO2_FPGA(void *Input, void *Output, void *Weight, void *Beta, int IFM_num, int OFM_num,
                    int Ksize, int Kstride, int Input_w, int Input_h, int Output_w, int Output_h,
                    int Padding, int IsNL, int IsBN, int TM, int TN, int TR, int TC,
                    int OFM_num_bound, int mLoopsxTM, int mLoops_a1xTM, int LayerType,
                    int Qw, int Qa_in, int Qa_out, int Qb)
{
    YOLO2_FPGA((IO_Dtype *)Input, (IO_Dtype *)Output, (IO_Dtype *)Weight, (IO_Dtype *)Beta, IFM_num,
               OFM_num, Ksize, Kstride, Input_w, Input_h, Output_w, Output_h, Padding, IsNL != 0,
               IsBN != 0, TM, TN, TR, TC, OFM_num_bound, mLoopsxTM, mLoops_a1xTM, LayerType, Qw, Qa_in,
               Qa_out, Qb);
}

// Whole network through the reference's own wrapper.  The reference reads weights/*.bin
// relative to the current directory, so the caller chdir()s into a scratch dir first.
// region_proc receives net->layers[n-1].output (71,825 floats, after forward_region_layer).
// The raw (pre-activation) tensor is written by the reference itself to the file named by
// YOLO2_DUMP_REGION_RAW_CPU.
int ref_yolov2_hls_ps(const char *cfg_path, const float *input, float *region_proc)
{
    try {
        network *net = load_network(const_cast<char *>(cfg_path));
        if (!net) return -1;
        set_batch_network(net, 1);
#ifdef INT16_MODE
        yolov2_hls_ps(net, input, Precision::INT16);
#else
        yolov2_hls_ps(net, input, Precision::FP32);
#endif
        layer last = net->layers[net->n - 1];
        std::memcpy(region_proc, last.output, sizeof(float) * last.outputs);
        return last.outputs;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ref_yolov2_hls_ps: %s\n", e.what());
        return -2;
    }
}

}  // extern "C"
