raise SystemExit('use `python -m sg2im_amd.build` to build the HIP library')
