#!/bin/bash
# One-shot probe of the GPU box for any OpenCV.  Output -> gpurun_out/r5_opencv_probe.txt
out=gpurun_out/r5_opencv_probe.txt
{
echo "== python cv2"; python3 -c "import cv2; print(cv2.__version__); print(cv2.getBuildInformation()[:600])" 2>&1
echo "== headers"; ls -d /usr/include/opencv4 /usr/local/include/opencv4 /usr/include/opencv2 /opt/*/include/opencv4 2>&1
echo "== ldconfig"; ldconfig -p 2>/dev/null | grep -i opencv
echo "== pip"; python3 -m pip list 2>/dev/null | grep -i -E "opencv|cv2|scikit-image|imageio|pillow|kornia|torchvision"
echo "== find"; find / -xdev \( -name 'cv2*.so' -o -name 'libopencv_core*' -o -name 'opencv*.pc' -o -name 'OpenCVConfig.cmake' \) 2>/dev/null | head -20
echo "== pkg-config"; pkg-config --modversion opencv4 2>&1
echo "== other image libs"; python3 -c "
import importlib
for m in ['PIL','skimage','imageio','torchvision','kornia','scipy.ndimage','mahotas','SimpleITK']:
    try:
        mod=importlib.import_module(m); print(m, getattr(mod,'__version__','?'))
    except Exception as e: print(m,'ABSENT',type(e).__name__)
"
echo "== nproc/mem"; nproc; free -g | head -2
echo "== done"
} > $out 2>&1
cat $out
exit 0
