/* _gsrcall: the drop-in packages' hop from Python into the C ABI (include/gsr.h) as a plain CPython extension.
 *
 * What it replaces: the ctypes route of gs_localization_amd/rasterizer.py -- 34 / 40 struct-field assignments through ctypes
 * descriptors plus a foreign call per forward / backward, ~10-15 us of interpreter time per call on the path the reference's
 * unchanged scripts take (diff_gaussian_rasterization[_pose].GaussianRasterizer -> here -> gsr_forward_packed / gsr_backward_packed;
 * call site served: gs_localization/pipelines/tools/__init__.py:130-141).  No torch headers: the caller hands over tensor
 * data_ptr()s as Python ints (None = NULL), so the C ABI stays the boundary; the module links libgsr_hip.so.
 *
 *   forward(state_addr, P, D, M, background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
 *           rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth,
 *           out_alpha, radii, debug, n_touched, stream, geom_ptr, geom_cap, bin_ptr, bin_cap, img_ptr, img_cap) -> int
 *       = gsr_forward_packed with the three workspaces handed over as fixed buffers (gsr_fixed_buffer_resize); the return value is
 *         the library's (num_rendered >= 0, or a negative GSR_E_* -- GSR_E_ALLOC when a workspace was too small).
 *   backward(P, D, M, R, background, width, height, means3D, shs, colors_precomp, alphas, scales, scale_modifier, rotations,
 *            cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom, binning, img, dL_dpix, dL_ddepths,
 *            dL_dalphas, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug,
 *            pose_mode, dL_dtau, stream) -> int
 *       = gsr_backward_packed.
 * The GIL is released around the library call (the forward of the stateless entry points ends with a blocking read). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include "gsr.h"

static int as_ptr(PyObject* o, void** out)
{
    if (o == Py_None) { *out = NULL; return 0; }
    *out = PyLong_AsVoidPtr(o);
    return (*out == NULL && PyErr_Occurred()) ? -1 : 0;
}
static int as_int(PyObject* o, int* out)
{
    const long v = PyLong_AsLong(o);
    if (v == -1 && PyErr_Occurred()) return -1;
    *out = (int)v;
    return 0;
}
static int as_size(PyObject* o, size_t* out)
{
    const size_t v = PyLong_AsSize_t(o);
    if (v == (size_t)-1 && PyErr_Occurred()) return -1;
    *out = v;
    return 0;
}
static int as_float(PyObject* o, float* out)
{
    const double v = PyFloat_AsDouble(o);
    if (v == -1.0 && PyErr_Occurred()) return -1;
    *out = (float)v;
    return 0;
}
#define PTR(i, dst) do { void* p_; if (as_ptr(args[i], &p_) < 0) return NULL; (dst) = p_; } while (0)
#define INT(i, dst) do { if (as_int(args[i], &(dst)) < 0) return NULL; } while (0)
#define FLT(i, dst) do { if (as_float(args[i], &(dst)) < 0) return NULL; } while (0)
#define SIZ(i, dst) do { if (as_size(args[i], &(dst)) < 0) return NULL; } while (0)

static PyObject* gsrcall_forward(PyObject* self, PyObject* const* args, Py_ssize_t nargs)
{
    (void)self;
    if (nargs != 34) { PyErr_SetString(PyExc_TypeError, "_gsrcall.forward takes 34 positional arguments"); return NULL; }
    gsr_forward_args a;
    gsr_fixed_buffer fb[3];
    PTR(0, a.state);
    INT(1, a.P); INT(2, a.D); INT(3, a.M);
    PTR(4, a.background);
    INT(5, a.width); INT(6, a.height);
    PTR(7, a.means3D); PTR(8, a.shs); PTR(9, a.colors_precomp); PTR(10, a.opacities);
    PTR(11, a.scales); FLT(12, a.scale_modifier); PTR(13, a.rotations); PTR(14, a.cov3D_precomp);
    PTR(15, a.viewmatrix); PTR(16, a.projmatrix); PTR(17, a.cam_pos);
    FLT(18, a.tan_fovx); FLT(19, a.tan_fovy);
    INT(20, a.prefiltered);
    PTR(21, a.out_color); PTR(22, a.out_depth); PTR(23, a.out_alpha);
    PTR(24, a.radii);
    INT(25, a.debug);
    PTR(26, a.n_touched);
    PTR(27, a.stream);
    for (int k = 0; k < 3; k++) {
        PTR(28 + 2 * k, fb[k].ptr);
        SIZ(29 + 2 * k, fb[k].capacity);
        fb[k].requested = 0;
    }
    a.geometry_buffer = a.binning_buffer = a.image_buffer = gsr_fixed_buffer_resize;
    a.geometry_ctx = &fb[0]; a.binning_ctx = &fb[1]; a.image_ctx = &fb[2];
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = gsr_forward_packed(&a);
    Py_END_ALLOW_THREADS
    return PyLong_FromLong(rc);
}

static PyObject* gsrcall_backward(PyObject* self, PyObject* const* args, Py_ssize_t nargs)
{
    (void)self;
    if (nargs != 40) { PyErr_SetString(PyExc_TypeError, "_gsrcall.backward takes 40 positional arguments"); return NULL; }
    gsr_backward_args b;
    INT(0, b.P); INT(1, b.D); INT(2, b.M); INT(3, b.R);
    PTR(4, b.background);
    INT(5, b.width); INT(6, b.height);
    PTR(7, b.means3D); PTR(8, b.shs); PTR(9, b.colors_precomp); PTR(10, b.alphas);
    PTR(11, b.scales); FLT(12, b.scale_modifier); PTR(13, b.rotations); PTR(14, b.cov3D_precomp);
    PTR(15, b.viewmatrix); PTR(16, b.projmatrix); PTR(17, b.campos);
    FLT(18, b.tan_fovx); FLT(19, b.tan_fovy);
    PTR(20, b.radii);
    PTR(21, b.geom_buffer); PTR(22, b.binning_buffer); PTR(23, b.img_buffer);
    PTR(24, b.dL_dpix); PTR(25, b.dL_ddepths); PTR(26, b.dL_dalphas);
    PTR(27, b.dL_dmean2D); PTR(28, b.dL_dconic); PTR(29, b.dL_dopacity); PTR(30, b.dL_dcolor);
    PTR(31, b.dL_dmean3D); PTR(32, b.dL_dcov3D); PTR(33, b.dL_dsh); PTR(34, b.dL_dscale); PTR(35, b.dL_drot);
    INT(36, b.debug);
    INT(37, b.pose_mode);
    PTR(38, b.dL_dtau);
    PTR(39, b.stream);
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = gsr_backward_packed(&b);
    Py_END_ALLOW_THREADS
    return PyLong_FromLong(rc);
}

static PyObject* gsrcall_last_error(PyObject* self, PyObject* noargs)
{
    (void)self; (void)noargs;
    return PyUnicode_FromString(gsr_last_error());
}

static PyMethodDef methods[] = {
    {"forward", (PyCFunction)(void (*)(void))gsrcall_forward, METH_FASTCALL, "gsr_forward_packed with fixed workspaces; returns the library's status"},
    {"backward", (PyCFunction)(void (*)(void))gsrcall_backward, METH_FASTCALL, "gsr_backward_packed; returns the library's status"},
    {"last_error", gsrcall_last_error, METH_NOARGS, "gsr_last_error() of the calling thread"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_gsrcall", "C-ABI hop of the drop-in rasterizer packages (include/gsr.h)", -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__gsrcall(void)
{
    PyObject* m = PyModule_Create(&moduledef);
    if (m == NULL) return NULL;
    if (PyModule_AddIntConstant(m, "ABI_VERSION", GSR_ABI_VERSION) < 0 || PyModule_AddIntConstant(m, "E_ALLOC", GSR_E_ALLOC) < 0) { Py_DECREF(m); return NULL; }
    return m;
}
