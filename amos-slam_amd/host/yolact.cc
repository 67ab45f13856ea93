// yolact.cc -- see yolact.h.  Reference: src/yolact.cc.
#include "yolact.h"

#include <Python.h>
#include <unistd.h>

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

bool yolact::evalImage(const cv::Mat &inputImage, cv::Mat &confidenceImage)
{
    if (!isInitializedResult()) return false;
    if (inputImage.empty() || inputImage.elemSize() != 1) {
        mstrErrDescription = "src image is empty!";
        return false;
    }
    const int h = inputImage.rows, w3 = inputImage.cols;  // 8UC3 frames arrive as rows x (3*cols) bytes in the stand-in Mat
    Gil gil;
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
