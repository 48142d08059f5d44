# round-5 GPU call R: the fft size fitted to the rows (FlashFFTConv._fit_seqlen) -- its tests, the tests whose shapes it re-routes, the reference's
# own test file at fft 16384 on the final library (the folded forward of call O became the default after the last full verbatim run), A/B timing
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_r; mkdir -p $O
cd $R
( timeout 150 python -m pytest tests/test_flashfftconv_gpu.py -q -x -k "fitted or cfg4 or levels_take or (unit_scale and big)" ) > $O/pytest_fit.txt 2>&1; tail -3 $O/pytest_fit.txt
( FFC_REF_TESTS_K=16384 timeout 120 python -m pytest tests/test_reference_verbatim_gpu.py -q -s -k flashfftconv ) > $O/verbatim_16384.txt 2>&1; tail -3 $O/verbatim_16384.txt
timeout 90 python benchmarks/fit_fft_ab.py 2>&1 | grep -v amdgpu.ids > $O/fit_fft_ab.txt; cat $O/fit_fft_ab.txt
( timeout 60 python -m pytest tests/test_graph_gpu.py -q -s ) 2>&1 | tail -4 > $O/graph.txt; cat $O/graph.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
