python profiles/epoch_cprofile.py 2>&1 | grep -v amdgpu.ids | cut -c1-150 | sed -n 3,40p
