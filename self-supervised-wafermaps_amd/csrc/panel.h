// Internal interface of panel.hip (short-reduction Linear layers), used by the Linear entry points in conv.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct WmPanelArgs {
  const uint16_t* x;       // [rows][192]
  const uint16_t* w;       // [N][192]: row n holds the coefficients of output column n
  const float* bias;       // [N] or NULL
  const uint16_t* res;     // [rows][N] or NULL (act 0)
  const uint16_t* pre_in;  // [rows][N] (act 2)
  uint16_t* pre_out;       // [rows][N] (act 1)
  uint16_t* y;             // [rows][N]
  int rows, N;
  int act;                 // 0: (+ bias) (+ residual); 1: + bias, pre_out, GELU; 2: * gelu'(pre_in)
  int tiles_per_block;     // set by wm_panel_launch
  const float* ln_gamma;   // optional LayerNorm applied to the token rows first (on the register fragments): [192]
  const float* ln_beta;
  float ln_eps;
  int debug;               // WM_PANEL_DEBUG bits (timing experiments): 1 no stores, 2 no MFMAs, 4 no staging, 8 no x rows
};

bool wm_panel_ok(long long rows, int C, int N, bool has_aux);
int wm_panel_launch(WmPanelArgs a, hipStream_t st);
