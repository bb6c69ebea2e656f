// pyutil.h -- Python dict / NumPy access for the runtime (the role of the reference's
// Kernel/src/convnet/util.cu:33-103 and the NumPy-view half of Kernel/src/matrix/matrix.cpp:71-90).
// Every function here must be called with the GIL held.
#pragma once
#include <Python.h>

#include <string>
#include <vector>

namespace rt {

typedef std::vector<int> intv;
typedef std::vector<float> floatv;

// Non-owning view of a 2-D float32 ndarray.  C-contiguous arrays are seen as (rows, cols)
// row-major; Fortran-contiguous ones set `fortran` and their bytes are (cols, rows) row-major.
// Anything else (other dtype, strided) is replaced by a C-contiguous float32 copy, as the
// reference does.  Unlike the reference, the view holds a reference to the array, so the memory
// stays valid while a worker thread uses it; release() must run with the GIL held.
class HostMat {
public:
    HostMat() {}
    explicit HostMat(PyObject* ndarray);   // throws std::runtime_error on a non-array / non-2-D object
    HostMat(const HostMat&) = delete;
    HostMat& operator=(const HostMat&) = delete;
    ~HostMat() {}                          // does NOT drop the Python reference (may run without the GIL)
    void release();                        // Py_DECREF; GIL required

    float* data() const { return _data; }
    int rows() const { return _rows; }
    int cols() const { return _cols; }
    bool fortran() const { return _fortran; }
    size_t numel() const { return (size_t)_rows * _cols; }
    PyObject* object() const { return _obj; }

    // Copies to / from a dense row-major (rows, cols) buffer regardless of the array's order.
    void toRowMajor(float* dst) const;
    void fromRowMajor(const float* src);

private:
    PyObject* _obj = nullptr;
    float* _data = nullptr;
    int _rows = 0, _cols = 0;
    bool _fortran = false;
};

typedef std::vector<HostMat*> HostMatV;

bool pyDictHas(PyObject* dict, const char* key);
int pyDictGetInt(PyObject* dict, const char* key);          // ints and bools
float pyDictGetFloat(PyObject* dict, const char* key);
std::string pyDictGetString(PyObject* dict, const char* key);
intv* pyDictGetIntV(PyObject* dict, const char* key);        // NULL when the key is absent
floatv* pyDictGetFloatV(PyObject* dict, const char* key);
HostMat* pyDictGetMatrix(PyObject* dict, const char* key);
HostMatV* pyDictGetMatrixV(PyObject* dict, const char* key);
HostMatV* getMatrixV(PyObject* list);
intv* getIntV(PyObject* list);

void initNumpy();   // import_array(), once, from module init

}  // namespace rt
