// yolact.cc -- see yolact.h.  Reference: src/yolact.cc.
#include "yolact.h"

#include <Python.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <sstream>

namespace ORB_SLAM2
{

namespace
{
struct Gil {
    PyGILState_STATE s;
    Gil() : s(PyGILState_Ensure()) {}
    ~Gil() { PyGILState_Release(s); }
};
}  // namespace

yolact::yolact(const std::string &pyFilePath, const std::string &modelPath, const size_t &categories)
    : mpTracker(nullptr), mbNewImgFlag(false), mSkipIndex(0), imgIndex(0), mpPyEvalModule(nullptr), mpPyEvalFunc(nullptr),
      mpPySessionEval(nullptr), mpSessionFrame(nullptr), mpSessionMask(nullptr), mnSessionH(0), mnSessionW(0), mnSessionMaskRows(0),
      mnSessionMaskCols(0), mbSessionUnavailable(false),
      mnCategories(categories), mbIsLEDNETInitializedOK(false), mbIsPythonInitializedOK(false), mbOwnsInterpreter(false),
      mbFinishRequested(false)
{
    if (!parseFilePathAndName(pyFilePath)) return;
    if (!Py_IsInitialized()) {
        Py_Initialize();
        if (!Py_IsInitialized()) {
            mstrErrDescription = "Python Env initialize failed.";
            return;
        }
        mbOwnsInterpreter = true;
        PyEval_SaveThread();  // let any thread take the GIL through PyGILState_Ensure
    }
    mbIsPythonInitializedOK = true;
    Gil gil;
    PyObject *sysPath = PySys_GetObject("path");  // borrowed
    PyObject *dir = PyUnicode_FromString(mstrPyMoudlePath.c_str());
    if (!sysPath || !dir || PyList_Insert(sysPath, 0, dir) != 0) {
        Py_XDECREF(dir);
        FetchPythonError("sys.path.insert('" + mstrPyMoudlePath + "') failed");
        return;
    }
    Py_DECREF(dir);
    PyObject *mod = PyImport_ImportModule(mstrPyMoudleName.c_str());
    if (!mod) {
        FetchPythonError("Error: import py moudle " + mstrPyMoudleName + " failed.");
        return;
    }
    mpPyEvalModule = mod;
    PyObject *init = PyObject_GetAttrString(mod, INIT_PY_FUNCTION_NAME);
    if (!init) {
        FetchPythonError(std::string("Error: funtc named \"") + INIT_PY_FUNCTION_NAME + "\" in py moudle \"" + mstrPyMoudleName + "\" not found.");
        return;
    }
    PyObject *ret = PyObject_CallFunction(init, "si", modelPath.c_str(), (int)categories);
    Py_DECREF(init);
    if (!ret) {
        FetchPythonError(std::string("Error occured when calling method \"") + INIT_PY_FUNCTION_NAME + "\"");
        return;
    }
    Py_DECREF(ret);
    mbIsLEDNETInitializedOK = true;
}

yolact::~yolact()
{
    if (mbIsPythonInitializedOK && Py_IsInitialized()) {
        Gil gil;
        Py_XDECREF((PyObject *)mpPyEvalFunc);
        Py_XDECREF((PyObject *)mpPySessionEval);
        Py_XDECREF((PyObject *)mpPyEvalModule);
    }
    // the interpreter is left alive: other embedders (and the reference) never finalize it either
}

bool yolact::parseFilePathAndName(const std::string &s)
{
    const size_t slash = s.find_last_of('/');
    const std::string file = slash == std::string::npos ? s : s.substr(slash + 1);
    mstrPyMoudlePath = slash == std::string::npos ? "." : s.substr(0, slash);
    const size_t dot = file.rfind(".py");
    if (file.empty() || dot == std::string::npos || dot + 3 != file.size()) {
        mstrErrDescription = "python file name \"" + s + "\" is not a .py file";
        return false;
    }
    mstrPyMoudleName = file.substr(0, dot);
    return true;
}

void yolact::FetchPythonError(const std::string &context)
{
    std::ostringstream ss;
    ss << context;
    if (PyErr_Occurred()) {
        PyObject *type = nullptr, *value = nullptr, *tb = nullptr;
        PyErr_Fetch(&type, &value, &tb);
        PyErr_NormalizeException(&type, &value, &tb);
        PyObject *str = value ? PyObject_Str(value) : nullptr;
        if (str) {
            const char *msg = PyUnicode_AsUTF8(str);
            if (msg) ss << " [" << msg << "]";
        }
        Py_XDECREF(str);
        Py_XDECREF(type);
        Py_XDECREF(value);
        Py_XDECREF(tb);
    }
    mstrErrDescription = ss.str();
}

// The GPU path: the frame goes straight into the session's pinned buffer, one Python call replays the captured graph, the mask is cloned
// out of the pinned mask buffer.  handled = false when the module has no session for this engine (CPU engine, AMOS_MASK_GRAPH=0, an
// older module): the caller then takes the bytes path.  Called with the GIL held.
bool yolact::evalThroughSession(const cv::Mat &inputImage, cv::Mat &confidenceImage, bool &handled)
{
    handled = false;
    if (mbSessionUnavailable) return false;
    const int h = inputImage.rows, w3 = inputImage.cols * (int)inputImage.elemSize(), w = w3 / 3;  // bytes per row: 8UC3, or rows x (3 * cols) of 8UC1
    {
        // asked for on every frame (a dictionary hit on the Python side): the buffers belong to the engine the module holds NOW -- another
        // yolact object's yolact_init may have replaced it since the last frame
        PyObject *fn = PyObject_GetAttrString((PyObject *)mpPyEvalModule, "yolact_frame_session");
        if (!fn) { PyErr_Clear(); mbSessionUnavailable = true; return false; }
        PyObject *ret = PyObject_CallFunction(fn, "ii", h, w);
        Py_DECREF(fn);
        if (!ret) { handled = true; FetchPythonError("Error occured when calling method \"yolact_frame_session\""); return false; }
        if (ret == Py_None) { Py_DECREF(ret); mbSessionUnavailable = true; return false; }
        unsigned long long in = 0, out = 0;
        int rows = 0, cols = 0;
        const bool ok = PyArg_ParseTuple(ret, "KKii", &in, &out, &rows, &cols) != 0;
        Py_DECREF(ret);
        if (!ok || !in || !out || rows < 1 || cols < 1) { handled = true; FetchPythonError("yolact_frame_session did not return (frame address, mask address, rows, columns)"); return false; }
        mpSessionFrame = reinterpret_cast<unsigned char *>((uintptr_t)in);
        mpSessionMask = reinterpret_cast<unsigned char *>((uintptr_t)out);
        mnSessionH = h; mnSessionW = w; mnSessionMaskRows = rows; mnSessionMaskCols = cols;
        if (!mpPySessionEval) mpPySessionEval = PyObject_GetAttrString((PyObject *)mpPyEvalModule, "yolact_eval_session");
        if (!mpPySessionEval) { handled = true; mpSessionFrame = nullptr; FetchPythonError("Error: function \"yolact_eval_session\" NOT found."); return false; }
    }
    handled = true;
    for (int y = 0; y < h; y++) std::memcpy(mpSessionFrame + (size_t)y * w3, inputImage.ptr(y), (size_t)w3);
    PyObject *ret = PyObject_CallFunction((PyObject *)mpPySessionEval, "ii", h, w);
    if (!ret) {
        FetchPythonError(std::string("Error occured when calling method \"") + EVAL_PY_FUNCTION_NAME + "\" in python module \"" + mstrPyMoudleName + "\". ");
        return false;
    }
    const int found = PyObject_IsTrue(ret);
    Py_DECREF(ret);
    if (found != 1) {  // the reference's Python raises IndexError here (masks[0] of an empty result) and evalImage returns false
        mstrErrDescription = std::string("Error occured when calling method \"") + EVAL_PY_FUNCTION_NAME + "\" in python module \"" + mstrPyMoudleName +
                             "\".  [no detection above the score threshold]";
        return false;
    }
    cv::Mat m(mnSessionMaskRows, mnSessionMaskCols, CV_8UC1, mpSessionMask);
    confidenceImage = m.clone();  // yolact.cc:315
    return true;
}

bool yolact::evalImage(const cv::Mat &inputImage, cv::Mat &confidenceImage)
{
    if (!isInitializedResult()) return false;
    if (inputImage.empty() || (inputImage.elemSize() != 1 && inputImage.elemSize() != 3)) {
        mstrErrDescription = "src image is empty!";
        return false;
    }
    // bytes per row: an 8UC3 frame (real OpenCV), or rows x (3 * cols) bytes of 8UC1 (the stand-in Mat has no channels)
    const int h = inputImage.rows, w3 = inputImage.cols * (int)inputImage.elemSize();
    Gil gil;
    {
        bool handled = false;
        const bool ok = evalThroughSession(inputImage, confidenceImage, handled);
        if (handled) return ok;
    }
    if (!mpPyEvalFunc) {
        // the fused entry point (raw BGR bytes; marshalling on the GPU) of mask/yolact_interface.py
        mpPyEvalFunc = PyObject_GetAttrString((PyObject *)mpPyEvalModule, "yolact_eval_bgr_bytes");
        if (!mpPyEvalFunc) {
            FetchPythonError("Error: YOLACT function named \"yolact_eval_bgr_bytes\" in python module \"" + mstrPyMoudleName + "\" NOT found.");
            return false;
        }
    }
    PyObject *buf = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)h * w3);
    if (!buf) { FetchPythonError("allocating the frame buffer failed"); return false; }
    char *dst = PyBytes_AsString(buf);
    for (int y = 0; y < h; y++) std::memcpy(dst + (size_t)y * w3, inputImage.ptr(y), (size_t)w3);
    PyObject *ret = PyObject_CallFunction((PyObject *)mpPyEvalFunc, "Oii", buf, h, w3 / 3);
    Py_DECREF(buf);
    if (!ret) {
        FetchPythonError(std::string("Error occured when calling method \"") + EVAL_PY_FUNCTION_NAME + "\" in python module \"" + mstrPyMoudleName + "\". ");
        return false;
    }
    bool ok = false;
    if (!PyTuple_Check(ret) || PyTuple_Size(ret) < 1) {
        mstrErrDescription = "Eval image function did NOT return a tuple.";
    } else {
        Py_buffer view;
        if (PyObject_GetBuffer(PyTuple_GetItem(ret, 0), &view, PyBUF_C_CONTIGUOUS | PyBUF_ND) != 0) {
            FetchPythonError("mask is not a contiguous buffer");
        } else {
            if (view.ndim == 2 && view.itemsize == 1) {
                cv::Mat m((int)view.shape[0], (int)view.shape[1], CV_8UC1, view.buf);
                confidenceImage = m.clone();
                ok = true;
            } else {
                mstrErrDescription = "mask must be a 2-D uint8 array";
            }
            PyBuffer_Release(&view);
        }
    }
    Py_DECREF(ret);
    return ok;
}

// ---- worker-thread members, yolact.cc:123-201 -------------------------------------------------
bool yolact::isNewImgArrived()
{
    std::unique_lock<std::mutex> lock(mMutexGetNewImg);
    if (mbNewImgFlag) {
        mbNewImgFlag = false;
        return true;
    }
    return false;
}

void yolact::ProduceImgSegment()
{
    std::unique_lock<std::mutex> lock(mMutexNewImgSegment);
    cv::Mat mask;
    if (evalImage(mImg, mask)) mMask = mask;
    imgIndex++;
}

void yolact::Run()
{
    while (!mbFinishRequested) {
        usleep(1);
        if (!isNewImgArrived()) continue;
        ProduceImgSegment();
    }
}

}  // namespace ORB_SLAM2
