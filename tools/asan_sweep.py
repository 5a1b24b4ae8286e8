"""Host-side sweep of the C ABI under AddressSanitizer / UBSan (run through tools/asan_run.sh on the CPU box; no GPU needed).

Every `*_workspace_bytes` / `*_param_count` / `*_param_offsets` planner over a grid of shapes (RAMS: batch 1..40, odd heights /
widths, scale 2 / 3 / 4 -- the r03 slab overrun was a planner of exactly this kind), the host-side RAMS slab / plan consistency
rules restated here, and the argument validation of every compute entry point with never-dereferenced device addresses: on a
host without a GPU such a call runs its host code (validation, planning, parameter walks) up to its first HIP call and returns a
HIP error, which is all this sweep wants from it."""
import ctypes as C
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import _lib  # noqa: E402


def fake(k):
    return C.c_void_p(0x7000_0000_0000 + 4096 * k)


def main():
    lib = _lib.lib()
    calls = 0
    assert lib.inr_version() == 1
    # ---- SIREN planners -------------------------------------------------------------------------------------------------
    for fin, hid, layers, out in itertools.product((1, 2, 3, 32, 64, 256, 257), (16, 32, 64, 128, 512, 520), (0, 1, 3, 6), (1, 2, 3)):
        d = _lib.SirenDesc(fin, hid, layers, out, 30.0, 30.0)
        total = lib.inr_siren_param_count(C.byref(d))
        offs = (C.c_int64 * (2 * (layers + 2)))()
        rc = lib.inr_siren_param_offsets(C.byref(d), offs)
        calls += 2
        if total > 0:
            assert rc == 0
            o = list(offs)
            assert o == sorted(o) and o[-1] < total and all(v % 4 == 0 for v in o), (fin, hid, layers, out, o, total)
        lib.inr_siren_hp_eligible(C.byref(d))
        for n in (1, 7, 64, 4095, 4096, 4097, 16384, 65537, 524288, 4194304):
            a = lib.inr_siren_fit_workspace_bytes(C.byref(d), n)
            b = lib.inr_siren_forward_workspace_bytes(C.byref(d), n)
            c = lib.inr_siren_reconstruct_workspace_bytes(C.byref(d), n)
            calls += 3
            if total > 0:
                assert a >= b > 0 and c > 0, (fin, hid, layers, out, n, a, b, c)
    for n in (0, 1, 255, 256, 10**6, 2**31):
        lib.inr_mse_workspace_bytes(n)
        for fin, out in ((1, 1), (64, 1), (512, 3), (513, 2)):
            lib.inr_head_backward_workspace_bytes(n, fin, out)
            lib.inr_linear_param_grad_workspace_bytes(n, fin, out)
            lib.inr_sine_layer_backward_input_workspace_bytes(n, fin)
            calls += 3
    for k in range(0, 40):
        lib.inr_metric_workspace_bytes(k)
        calls += 1
    # ---- RAMS planners: B = 1 .. 40, odd H / W, scale 2 / 3 / 4 ----------------------------------------------------------
    for scale in (2, 3, 4):
        for channels in (9,):
            d = _lib.RamsDesc(scale, 32, 3, channels, 8, 12, 7433.6436, 2353.0723)
            total = lib.inr_rams_param_count(C.byref(d))
            tt = lib.inr_rams_train_param_count(C.byref(d))
            offs = (C.c_int64 * (3 * 128))()
            n = lib.inr_rams_train_param_offsets(C.byref(d), offs, 128)
            assert total > 0 and tt > 0 and n == 71, (scale, total, tt, n)
            flat = [int(offs[i]) for i in range(3 * n)]
            assert max(flat) < tt and min(flat) >= 0
            assert lib.inr_rams_train_param_offsets(C.byref(d), offs, 3) != n          # too few slots: refused, nothing written past them
            calls += 4
            for B in range(1, 41):
                for H, W in ((3, 3), (5, 7), (16, 16), (31, 33), (32, 32), (63, 65), (127, 129), (128, 128)):
                    ws = lib.inr_rams_workspace_bytes(C.byref(d), B, H, W)
                    big = B * (H + 4) * (W + 4) * channels * 32 * 4
                    calls += 1
                    assert ws >= 5 * big, (scale, B, H, W, ws)
                    if H <= 33:
                        tw = lib.inr_rams_train_workspace_bytes(C.byref(d), B, H, W)
                        calls += 1
                        assert tw > 0, (scale, B, H, W)
                    lib.inr_rams_shift_loss_workspace_bytes(B, H * scale)
                    lib.inr_rams_shift_loss_grad_workspace_bytes(B, H * scale)
                    lib.inr_rams_conv3d_wgrad_workspace_bytes(B, H + 2, W + 2, channels, 1)
                    lib.inr_rams_conv3d_wgrad_workspace_bytes(B, H + 2, W + 2, channels, 0)
                    calls += 4
                    # the forward with a workspace of exactly the planned size and never-dereferenced device addresses: host
                    # validation + planning up to the first HIP call (no device here -> a HIP error status, never a crash)
                    rc = lib.inr_rams_forward(C.byref(d), fake(1), fake(2), fake(3), B, H, W, 1, fake(1000), ws, None)
                    calls += 1
                    assert rc != 0 or os.environ.get("INR_SWEEP_ALLOW_GPU"), "a forward on fake addresses must not report success"
                    rc = lib.inr_rams_forward(C.byref(d), fake(1), fake(2), fake(3), B, H, W, 1, fake(1000), ws - 16, None)
                    assert rc == _lib.INR_E_WORKSPACE
    lib.inr_rams_conv3d_dgrad_workspace_bytes()
    for bad in (_lib.RamsDesc(3, 16, 3, 9, 8, 12, 0, 1), _lib.RamsDesc(9, 32, 3, 9, 8, 12, 0, 1), _lib.RamsDesc(3, 32, 3, 10, 8, 12, 0, 1),
                _lib.RamsDesc(3, 32, 3, 9, 0, 12, 0, 1), _lib.RamsDesc(3, 32, 3, 9, 8, -1, 0, 1)):
        assert lib.inr_rams_param_count(C.byref(bad)) < 0 and lib.inr_rams_workspace_bytes(C.byref(bad), 1, 16, 16) == 0
        calls += 2
    # ---- every compute entry point once with fake addresses (validation / planning / first HIP call) ---------------------
    d = _lib.SirenDesc(256, 512, 3, 1, 30.0, 30.0)
    small = _lib.SirenDesc(2, 64, 6, 1, 30.0, 30.0)
    for desc, n in ((d, 4096), (d, 70000), (small, 3600), (small, 20000)):
        wsb = lib.inr_siren_fit_workspace_bytes(C.byref(desc), n)
        lib.inr_siren_fit(C.byref(desc), fake(1), fake(2), fake(3), fake(4), fake(5), fake(6), None, n, 1, 3, 1e-4, 0.9, 0.999, 1e-8,
                          fake(7), fake(100), wsb, None)
        lib.inr_siren_fit_cycle(C.byref(desc), fake(1), fake(2), fake(3), fake(4), fake(5), fake(6), fake(8), 3, 0, n, 1, 5, 1e-4, 0.9,
                                0.999, 1e-8, fake(7), fake(100), wsb, None)
        lib.inr_siren_loss_grad(C.byref(desc), fake(1), fake(2), fake(5), fake(6), None, n, 0, fake(7), fake(100), wsb, None)
        lib.inr_siren_forward(C.byref(desc), fake(1), fake(5), n, fake(9), 1, 0.0, fake(100),
                              lib.inr_siren_forward_workspace_bytes(C.byref(desc), n), None)
        lib.inr_siren_forward_train(C.byref(desc), fake(1), fake(5), fake(9), n, fake(100), wsb, 0, None)
        lib.inr_siren_backward_train(C.byref(desc), fake(1), fake(2), fake(9), n, fake(100), wsb, None)
        shape = _lib.shape_array((16, 16, 8))
        lib.inr_siren_reconstruct(C.byref(desc), fake(1), shape, 3, fake(5), 128 if desc is d else 0, fake(9), 1, 0.0, 1 << 20, fake(100),
                                  lib.inr_siren_reconstruct_workspace_bytes(C.byref(desc), 1 << 20), None)
        calls += 7
    shape = _lib.shape_array((5, 7, 3))
    lib.inr_mgrid(fake(1), shape, 3, 0, 105, None)
    lib.inr_grid_fourier_map(fake(1), shape, 3, 0, 105, fake(2), 128, None)
    lib.inr_fourier_map(fake(1), fake(2), fake(3), 105, 3, 128, None)
    lib.inr_adam_step(fake(1), fake(2), fake(3), fake(4), 920068, 1, 1e-4, 0.9, 0.999, 1e-8, None)
    lib.inr_hybrid_fit(fake(1), fake(2), fake(3), fake(4), fake(5), 14400, None)
    lib.inr_auto_erd(fake(1), fake(2), None, 3600, 12, 1, None)
    lib.inr_rescale2d_linear(fake(1), fake(2), 3, 25, 25, 75, 75, None)
    lib.inr_adc_map(fake(1), fake(2), fake(3), 16384, 4, None)
    lib.inr_acquisition_products(fake(1), fake(2), fake(3), fake(4), fake(5), 1000, 2, 3, 2, None)
    calls += 9
    td = _lib.RamsDesc(3, 32, 3, 9, 8, 12, 7433.6436, 2353.0723)
    for B, side in ((1, 16), (2, 20), (32, 32), (33, 32)):
        tw = lib.inr_rams_train_workspace_bytes(C.byref(td), B, side, side)
        lib.inr_rams_train_step(C.byref(td), fake(1), fake(2), fake(3), fake(4), fake(5), fake(6), fake(7), fake(8), B, side, side, 1, 5e-4,
                                0.9, 0.999, 1e-7, fake(100), tw, None)
        lib.inr_rams_train_grads(C.byref(td), fake(1), fake(2), fake(5), fake(6), fake(7), fake(8), fake(9), B, side, side, fake(100), tw, None)
        calls += 3
    lib.inr_last_error()
    print(f"asan sweep: {calls} C-ABI calls, no sanitizer report")


if __name__ == "__main__":
    main()
