import ctypes as C, os, sys
os.environ["LFAMD_NO_TORCH"]="1"
L = C.CDLL("llamafile_amd/libllamafile_amd_hip.so")
L.lfamd_last_error.restype = C.c_char_p
print("count", L.lfamd_device_count())
rc = L.lfamd_init(0)
print("init rc", rc, L.lfamd_last_error())
H = C.CDLL("llamafile_amd/libllamafile_sgemm.so")
H.llamafile_sgemm_amd_error.restype = C.c_char_p
print("host available", H.llamafile_sgemm_amd_available(), H.llamafile_sgemm_amd_error())
