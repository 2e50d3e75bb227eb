function reference_fixture(in_file, out_file, reference_root)
% REFERENCE_FIXTURE  run the REFERENCE solver (aldma/qpdo, MATLAB class + mex over CHOLMOD) on an instance written by
% qpdo_amd.io.save_mat and store its answer in the same file format, so that it can be dropped into tests/golden/ as
% ext_<name>.mat and checked by tests/test_gpu_ext.py (status / iterations / oterations identical, x and y within the
% stated fp64 tolerance).  This is the route by which genuine-CHOLMOD results can pin the parity tests.
%
%   reference_fixture('c1.mat', 'ext_c1.mat', '/path/to/qpdo')     % needs interfaces/mex/qpdo_mex.mex* built
%
% Follows the call sequence of examples/demo_mex.m:19-31 and the output marshalling of interfaces/mex/qpdo_mex.c:227-281.
if nargin >= 3, addpath(fullfile(reference_root, 'interfaces', 'mex')); end
S = load(in_file);
solver = qpdo;
settings = solver.default_settings();
if isfield(S, 'settings')
    f = fieldnames(S.settings);
    for k = 1:numel(f)
        if isfield(settings, f{k}), settings.(f{k}) = double(S.settings.(f{k})); end
    end
end
settings.verbose = 0;
solver.setup(S.Q, S.q(:), S.A, S.l(:), S.u(:), settings);
if isfield(S, 'x0') && isfield(S, 'y0'), solver.warm_start(S.x0(:), S.y0(:)); end
res = solver.solve();
ref = struct();
ref.x = res.x; ref.y = res.y;
ref.prim_inf_cert = res.prim_inf_cert; ref.dual_inf_cert = res.dual_inf_cert;
ref.status_val = double(res.info.status_val);
ref.iterations = double(res.info.iterations);
ref.oterations = double(res.info.oterations);
ref.res_prim_norm = res.info.res_prim_norm;
ref.res_dual_norm = res.info.res_dual_norm;
ref.objective = res.info.objective;
ref.source = 'aldma/qpdo reference (MATLAB mex, CHOLMOD)';
S.ref = ref;
S.settings = settings;
solver.delete();
save(out_file, '-struct', 'S', '-v7');     % v7 (not v7.3): readable by scipy.io.loadmat
end
