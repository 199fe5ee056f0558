"""CPU: host-side logic of the drop-in modules -- state-dict keys / parameter order / seeded init
identical to the reference (goldens captured by oracle/make_golden.py), SMPL model packing, and
the no-CPU-fallback rule."""
import json
import os

import numpy as np
import pytest
import torch

import straps_amd

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
MP = straps_amd.synthetic_mean_params(0)


@pytest.mark.parametrize('layers', [18, 50])
def test_state_dict_manifest_and_param_order(layers):
    man = json.load(open(os.path.join(GOLD, 'state_dict_keys_r%d.json' % layers)))
    m = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=MP)
    sd = m.state_dict()
    assert list(sd.keys()) == list(man['keys'].keys())
    assert all(list(sd[k].shape) == man['keys'][k] for k in sd)
    assert [n for n, _ in m.named_parameters()] == man['param_order']
    assert len(sd) == {18: 132, 50: 330}[layers]
    # duplicate registration: aliases share storage (models/ief_module.py:24-28)
    assert sd['ief_module.fc1.weight'].data_ptr() == sd['ief_module.ief_layers.0.weight'].data_ptr()


@pytest.mark.parametrize('layers', [18, 50])
def test_seeded_construction_matches_reference(layers):
    gold = json.load(open(os.path.join(GOLD, 'init_checksums.json')))['r%d' % layers]
    torch.manual_seed(1234)
    m = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=MP)
    sd = m.state_dict()
    for k, (s, a, head) in gold.items():
        v = sd[k]
        assert float(v.double().sum()) == pytest.approx(s, rel=1e-9, abs=1e-9), k
        assert float(v.double().abs().sum()) == pytest.approx(a, rel=1e-9, abs=1e-9), k
        assert [float(x) for x in v.reshape(-1)[:3]] == head, k


def test_load_state_dict_strict_roundtrip():
    m = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP)
    m2 = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP)
    m2.load_state_dict(m.state_dict(), strict=True)
    ck = {'epoch': 3, 'best_epoch': 2, 'best_epoch_val_metrics': {'pves_pa': np.float64(0.1)},
          'model_state_dict': m.state_dict(), 'best_model_state_dict': m.state_dict(),
          'optimiser_state_dict': torch.optim.Adam(m.parameters(), lr=1e-4).state_dict(), 'criterion_state_dict': {}}
    keys = json.load(open(os.path.join(GOLD, 'criterion_keys.json')))['checkpoint_keys']
    assert sorted(ck) == sorted(keys)


def test_no_cpu_fallback():
    m = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match='GPU tensor'):
        m(torch.zeros(1, 18, 256, 256))
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=2)
    with torch.no_grad(), pytest.raises(RuntimeError, match='GPU tensor'):
        smpl(betas=torch.zeros(2, 10))
    with pytest.raises(RuntimeError, match='GPU tensor'):
        straps_amd.rot6d_to_rotmat(torch.zeros(2, 144))


def test_missing_assets_raise_like_reference(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError):
        straps_amd.SingleInputRegressor(18, 18, 3)
    with pytest.raises(FileNotFoundError):
        straps_amd.SMPL('additional/smpl', batch_size=1)
    m = straps_amd.SingleInputRegressor(18, 34, 3, mean_params=MP)     # silently empty, fails late (regressor.py:28-41)
    with pytest.raises(AttributeError):
        m(torch.zeros(1, 18, 256, 256))


def test_ief_initial_estimate():
    m = straps_amd.IEFModule([512, 512], 512, 157, mean_params=MP)
    v = m.initial_params_estimate
    assert v.shape == (157,) and float(v[0]) == pytest.approx(0.9) and float(v[1]) == 0 and float(v[2]) == 0
    np.testing.assert_array_equal(v[3:147].numpy(), MP['pose'])
    np.testing.assert_array_equal(v[147:].numpy(), MP['shape'])
    assert 'initial_params_estimate' not in m.state_dict()


def test_smpl_model_packing_roundtrip():
    model = straps_amd.synthetic_smpl_model(0)
    pk = straps_amd.pack_smpl_model(model)
    frag = pk['blend_frag'].reshape(pk['n_tiles'], 3, 28, 2, 32, 4)
    # spot-check the fragment mapping D[8g+4h+e][32t+i][c]
    for (t, c, g, h, i, e) in [(0, 0, 0, 0, 0, 0), (5, 1, 3, 1, 7, 2), (215, 2, 27, 0, 9, 1), (100, 0, 1, 1, 31, 3)]:
        k, v = 8 * g + 4 * h + e, 32 * t + i
        if v >= 6890 or k >= 218:
            want = 0.0
        elif k == 0:
            want = model['v_template'][v, c]
        elif k <= 10:
            want = model['shapedirs'][v, c, k - 1]
        else:
            want = model['posedirs'][k - 11, v * 3 + c]
        assert frag[t, c, g, h, i, e] == np.float32(want)
    assert pk['skin_k'] == 4 and pk['max_depth'] == 8
    W = np.zeros((6890, 24), np.float32)
    np.add.at(W, (np.repeat(np.arange(6890), 4), pk['skin_j'][:6890].reshape(-1)), pk['skin_w'][:6890].reshape(-1))
    np.testing.assert_array_equal(W, model['weights'])
    # virtual vertices: skinned with the packed tables they reproduce J_regressor_* @ LBS(vertices) for random
    # blend features F and random bone transforms A (fp64 emulation of the kernel's arithmetic)
    nt = pk['n_tiles']
    assert nt % 8 == 0 and nt >= 216 and pk['vj_ptr'].shape == (46,)
    nvirt = int(pk['vj_ptr'][-1])
    assert 0 < nvirt <= (nt - 216) * 32
    rs = np.random.RandomState(3)
    F = np.zeros(224)
    F[0] = 1.0
    F[1:218] = rs.randn(217) * 0.3
    A = rs.randn(24, 3, 4)
    Dk = pk['blend_frag'].reshape(nt, 3, 28, 2, 32, 4).transpose(2, 3, 5, 0, 4, 1).reshape(224, nt * 32, 3).astype(np.float64)
    vp = np.einsum('k,kvc->vc', F, Dk)                                                   # [nt*32][3]
    T = np.einsum('vk,vkrc->vrc', pk['skin_w'].astype(np.float64), A[pk['skin_j']])       # [nt*32][3][4]
    out = np.einsum('vrc,vc->vr', T[:, :, :3], vp) + T[:, :, 3]
    want = np.concatenate([model['J_regressor_extra'], model['J_regressor_cocoplus'], model['J_regressor_h36m']]).astype(np.float64) @ out[:6890]
    got = np.stack([out[6912 + pk['vj_ptr'][j]:6912 + pk['vj_ptr'][j + 1]].sum(0) for j in range(45)])
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * np.abs(want).max())
    assert np.all(out[6912 + nvirt:] == 0)                                                 # zero padding skins to zero


def test_checkpoint_roundtrip_reference_schema(tmp_path):
    """the .tar dict of train loop :369-377 round-trips, loads with the reference's reader logic and
    restores a regressor strictly."""
    from straps_amd import checkpoint_utils as cu
    m = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP)
    crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D'])
    opt = torch.optim.Adam(list(m.parameters()) + list(crit.parameters()), lr=1e-4)
    path = str(tmp_path / 'straps_model_checkpoint_exp001_epoch10.tar')
    cu.save_checkpoint(path, 10, m, opt, crit, best_epoch=7, best_epoch_val_metrics={'pves_pa': np.float64(0.08), 'mpjpes_pa': np.float64(0.06)})
    ck = cu.load_checkpoint(path)
    assert tuple(sorted(ck)) == tuple(sorted(cu.CHECKPOINT_KEYS))
    cur, best_epoch, best_wts, best_metrics = cu.load_training_info_from_checkpoint(ck, ['pves_pa', 'new_metric'])
    assert cur == 11 and best_epoch == 7 and best_metrics == {'pves_pa': 0.08, 'new_metric': np.inf}
    m2 = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP)
    m2.load_state_dict(ck['best_model_state_dict'], strict=True)            # run_predict.py:15-16
    crit.load_state_dict(ck['criterion_state_dict'])
    opt.load_state_dict(ck['optimiser_state_dict'])                           # run_train.py:207-209
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    with pytest.raises(KeyError):
        torch.save({'epoch': 1}, path)
        cu.load_checkpoint(path)


def test_pmc_summary_and_bench_traffic_reader(tmp_path):
    """tools/pmc_summary.py --json turns rocprofv3 counter CSVs into per-kernel HBM bytes per launch (FETCH_SIZE in KiB with the
    gfx950 x2 correction, WRITE_SIZE in KiB) and bench.pmc_traffic() reports the launch-weighted mean for the dominant kernel."""
    import importlib.util, json, subprocess, sys, types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = ['Kernel_Name,Counter_Name,Counter_Value,Dispatch_Id,Start_Timestamp,End_Timestamp']
    for i, (fetch, write, name) in enumerate([(1000, 500, 'void (anonymous namespace)::conv_igemm_kernel<64, 64, 2>((anonymous namespace)::ConvP)'),
                                              (3000, 1500, 'void (anonymous namespace)::conv_igemm_kernel<64, 64, 2>((anonymous namespace)::ConvP)'),
                                              (8000, 4000, 'void (anonymous namespace)::conv_igemm_kernel<128, 128, 2>((anonymous namespace)::ConvP)'),
                                              (100, 100, '(anonymous namespace)::bn_apply_kernel(float const*)')]):
        rows.append('"%s",FETCH_SIZE,%d,%d,1000,3000' % (name, fetch, i))
        rows.append('"%s",WRITE_SIZE,%d,%d,1000,3000' % (name, write, i))
    csv_path = tmp_path / 'counter_collection.csv'
    csv_path.write_text('\n'.join(rows) + '\n')
    out = tmp_path / 'pmc_traffic.json'
    subprocess.run([sys.executable, os.path.join(root, 'tools', 'pmc_summary.py'), '--json', str(out), 'train_r18_b64', 'unit test', str(csv_path)],
                   check=True, capture_output=True)
    doc = json.load(open(out))['train_r18_b64']
    k = doc['kernels']['void conv_igemm_kernel<64, 64, 2>']
    assert k['launches'] == 2 and k['hbm_read_bytes'] == 2 * 1024 * 2000 and k['hbm_write_bytes'] == 1024 * 1000
    # the reader: launch-weighted mean over both tile variants of the dominant kernel
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    prof = os.path.join(root, 'profiles', 'pmc_traffic.json')
    committed = json.load(open(prof))
    def match(n, kernel):
        n = n.replace('void ', '')
        return n.startswith(kernel) or (kernel == 'conv_igemm_x3_kernel' and n.startswith('conv_igemm_x3h_kernel'))

    for tag, prec, kernel in (('train_r18_b64', 'bf16x3', 'conv_igemm_x3_kernel'), ('train_r18_b64_fp32conv', 'fp32', 'conv_igemm_kernel')):
        assert tag in committed and any(match(n, kernel) for n in committed[tag]['kernels']), tag
        got = bench.pmc_traffic(types.SimpleNamespace(workload='train', layers=18, batch=0, conv_precision=prec, smpl_precision='fp16x3_lbs'), kernel)
        ks = [v for n, v in committed[tag]['kernels'].items() if match(n, kernel)]
        want = sum(v['launches'] * (v['hbm_read_bytes'] + v['hbm_write_bytes']) for v in ks) / sum(v['launches'] for v in ks)
        assert got['traffic'] == round(want)
    assert bench.pmc_traffic(types.SimpleNamespace(workload='train', layers=101, batch=0, conv_precision='bf16x3', smpl_precision='fp16x3_lbs'),
                             'conv_igemm_x3_kernel') == {}


def write_real_layout_model(tmp_path, model):
    """write `model` the way the files the reference loads are laid out: <dir>/SMPL_NEUTRAL.pkl (models/smpl_official.py:15 ->
    smplx: pickle with latin1 strings, scipy-sparse J_regressor, posedirs [6890,3,207], kintree_table [2,24] whose root parent
    is 2^32-1, faces 'f' as uint32, 300-column shapedirs) + the three regressor .npy files of config.py:6-8."""
    import pickle
    import scipy.sparse as sp
    d = tmp_path / 'smpl'
    d.mkdir()
    parents = np.asarray(model['parents'], np.int64).copy()
    parents[0] = 2 ** 32 - 1
    sd300 = np.zeros((6890, 3, 300), np.float64)
    sd300[:, :, :10] = model['shapedirs']
    raw = {'v_template': model['v_template'].astype(np.float64), 'shapedirs': sd300,
           'posedirs': model['posedirs'].astype(np.float64).T.reshape(6890, 3, 207),
           'J_regressor': sp.csc_matrix(model['J_regressor'].astype(np.float64)), 'weights': model['weights'].astype(np.float64),
           'kintree_table': np.stack([parents, np.arange(24)]).astype(np.uint32), 'f': model['faces'].astype(np.uint32),
           'bs_type': 'lrotmin', 'bs_style': 'lbs'}
    with open(d / 'SMPL_NEUTRAL.pkl', 'wb') as f:
        pickle.dump(raw, f, protocol=2)
    extra = []
    for key, fn in (('J_regressor_extra', 'J_regressor_extra.npy'), ('J_regressor_cocoplus', 'cocoplus_regressor.npy'),
                    ('J_regressor_h36m', 'J_regressor_h36m.npy')):
        np.save(tmp_path / fn, model[key])
        extra.append(str(tmp_path / fn))
    return str(d), tuple(extra)


def test_load_smpl_model_from_real_file_layout(tmp_path, monkeypatch):
    """S0 loader (models/smpl_official.py:15-25, config.py:3-10) on a synthetic file in the real layout."""
    model = straps_amd.synthetic_smpl_model(0)
    d, extra = write_real_layout_model(tmp_path, model)
    got = straps_amd.load_smpl_model(d, 'neutral', extra)
    for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'weights', 'J_regressor_extra', 'J_regressor_cocoplus', 'J_regressor_h36m'):
        assert got[k].dtype == np.float32 and np.array_equal(got[k], model[k]), k
    assert got['shapedirs'].shape == (6890, 3, 10) and got['posedirs'].shape == (207, 20670)
    assert np.array_equal(got['parents'], model['parents']) and got['parents'][0] == -1
    assert np.array_equal(got['faces'], model['faces']) and np.array_equal(got['extra_vertex_ids'], model['extra_vertex_ids'])
    # the packed kernel tables are identical too (what the device sees)
    pa, pb = straps_amd.pack_smpl_model(got), straps_amd.pack_smpl_model(model)
    for k in pb:
        assert np.array_equal(np.asarray(pa[k]), np.asarray(pb[k])), k
    # default regressor locations are the reference's relative paths (cwd-relative, config.py:6-8)
    (tmp_path / 'additional').mkdir()
    for src, dst in zip(extra, ('J_regressor_extra.npy', 'cocoplus_regressor.npy', 'J_regressor_h36m.npy')):
        os.replace(src, tmp_path / 'additional' / dst)
    monkeypatch.chdir(tmp_path)
    again = straps_amd.load_smpl_model(d)
    assert np.array_equal(again['J_regressor_h36m'], model['J_regressor_h36m'])
    with pytest.raises(FileNotFoundError):
        straps_amd.load_smpl_model(str(tmp_path / 'nowhere'))
    os.remove(tmp_path / 'additional' / 'cocoplus_regressor.npy')
    with pytest.raises(FileNotFoundError):
        straps_amd.load_smpl_model(d)


def test_split_precision_blend_operand_error_budget():
    """the fp16 two-term split of the blend directions (straps_hip.h: blend_frag_h) and the three-product contraction the
    kernel runs, emulated in numpy: the blend result stays within 3e-6 m of the float64 contraction for realistic (and one extreme)
    (theta, beta) -- two orders inside north_star's 1e-4 -- and the fragment mapping is the documented one."""
    import straps_oracle as O
    model = straps_amd.synthetic_smpl_model(0)
    pk = straps_amd.pack_smpl_model(model)
    nt = pk['n_tiles']
    fh = pk['blend_frag_h'].reshape(nt, 14, 3, 2, 2, 32, 8)                  # [t][s][c][hi|lo][hh][i][j]
    assert fh.dtype == np.float16
    unscale = pk['blend_h_unscale']
    sd = 1.0 / (unscale * 64.0)
    assert sd == 2.0 ** round(np.log2(sd)) and sd <= 2.0 ** 14
    # D[k = 16s + 8hh + j][v = 32t + i][c] back from the fragments
    Dsplit = fh.astype(np.float64).transpose(3, 1, 4, 6, 0, 5, 2).reshape(2, 224, nt * 32, 3)
    D = np.zeros((224, nt * 32, 3))
    D[0, :6890] = model['v_template']
    D[1:11, :6890] = np.transpose(model['shapedirs'], (2, 0, 1))
    D[11:218, :6890] = model['posedirs'].reshape(207, 6890, 3)
    rec = (Dsplit[0] + Dsplit[1]) / sd
    err = np.abs(rec[:, :6890] - D[:, :6890])
    assert err.max() <= 2.0 ** -21 * np.abs(D).max() and (err <= np.abs(D[:, :6890]) * 2.0 ** -21 + 2.0 ** -24 / sd).all()
    # the kernel's arithmetic: F scaled by 64, split in fp16; Fh.Dh + Fh.Dl + Fl.Dh accumulated in fp32
    rs = np.random.RandomState(0)
    B = 16
    betas = rs.randn(B, 10) * 1.5
    betas[0] = 10.0                                                              # far outside the training distribution
    import torch
    R = O.batch_rodrigues(torch.from_numpy(rs.randn(B * 24, 3) * 0.3)).view(B, 24, 3, 3).numpy()
    F = np.zeros((B, 224), np.float32)
    F[:, 0] = 1.0
    F[:, 1:11] = betas
    F[:, 11:218] = (R[:, 1:] - np.eye(3)).reshape(B, 207)
    Fs = F * np.float32(64.0)
    Fhh = Fs.astype(np.float16)
    Fll = (Fs - Fhh.astype(np.float32)).astype(np.float16)
    Dh, Dl = Dsplit[0, :, :6890].reshape(224, -1), Dsplit[1, :, :6890].reshape(224, -1)
    acc = (Fhh.astype(np.float32) @ Dh.astype(np.float32) + Fhh.astype(np.float32) @ Dl.astype(np.float32)
           + Fll.astype(np.float32) @ Dh.astype(np.float32)) * np.float32(unscale)
    exact = F.astype(np.float64) @ D[:, :6890].reshape(224, -1)
    e = float(np.abs(acc - exact).max())
    e32 = float(np.abs((F @ D[:, :6890].reshape(224, -1).astype(np.float32)) - exact).max())
    print('split-precision blend: max abs error %.2e m (plain fp32 contraction: %.2e m)' % (e, e32))
    assert e < 3e-6 and e <= 1.5 * e32 + 1e-7 and np.isfinite(acc).all()      # no worse than a plain fp32 contraction


def test_k_packed_skinning_operand_reproduces_the_three_product_split():
    """skin_frag_p (smpl.pack_smpl_model; the 64-body SMPL kernel's B operand): the three products of the two-term fp16 split packed along K,
    T = [Ah | Al | Ah | -] . [Wh | Wh | Wl | 0] over 80 columns (5 k-steps of 16; columns 72..79 carry zero weights, so the A side may hold
    anything finite there).  Emulated in float64 from the packed array exactly as the kernel reads it -- lane (i, hh) of k-step s holds
    packed columns 16 s + 8 hh .. + 7, and k-steps 3 / 4 re-use the A registers of k-steps 0 / 1 -- it must equal Ah.Wh + Al.Wh + Ah.Wl, and
    that must be the fp32-class value of A.W."""
    import straps_amd
    from straps_amd.smpl import pack_smpl_model
    model = straps_amd.synthetic_smpl_model(0)
    packed = pack_smpl_model(model)
    n_tiles = packed['n_tiles']
    P = packed['skin_frag_p'].reshape(n_tiles, 5, 2, 32, 8).astype(np.float64)          # [tile][kstep][hh][i][j]
    Wd = np.zeros((n_tiles * 32, 24), np.float32)
    Wd[:6890] = np.asarray(model['weights'], np.float32)
    Ws = Wd * np.float32(2.0 ** 14)
    Wh16 = Ws.astype(np.float16)
    Wl16 = (Ws - Wh16.astype(np.float32)).astype(np.float16)
    Wh, Wl = Wh16.astype(np.float64), Wl16.astype(np.float64)
    # (rows 6890.. of the packed operand are the virtual vertices of the regressed joints: checked through the product below only for the mesh)
    assert np.array_equal(P.transpose(0, 3, 1, 2, 4).reshape(n_tiles * 32, 80)[:6890, :24], Wh[:6890])
    rng = np.random.default_rng(0)
    A = rng.uniform(-3, 3, (5, 24)).astype(np.float32)                                  # joint-transform entries of 5 bodies
    As = (A * np.float32(1024.0)).astype(np.float32)
    Ah = As.astype(np.float16)
    Al = (As - Ah.astype(np.float32)).astype(np.float16)
    Ah64, Al64 = Ah.astype(np.float64), Al.astype(np.float64)
    # the kernel's A operand per packed k-step: rows [Ah 24 | Al 24]; k-step s, half hh reads 8 halves at 16 s + 8 hh (s < 3); s = 3, 4 re-use 0, 1
    row = np.concatenate([Ah64, Al64], axis=1)                                          # [body][48]
    T = np.zeros((5, n_tiles * 32))
    Pv = P.transpose(0, 3, 1, 2, 4).reshape(n_tiles * 32, 5, 2, 8)                      # [vertex][kstep][hh][j]
    for s_ in range(5):
        src = s_ % 3 if s_ < 3 else s_ - 3
        for hh in range(2):
            a = row[:, 16 * src + 8 * hh: 16 * src + 8 * hh + 8]                        # [body][8]
            T += a @ Pv[:, s_, hh].T
    want = Ah64 @ Wh.T + Al64 @ Wh.T + Ah64 @ Wl.T
    assert np.array_equal(T[:, :6890], want[:, :6890])
    W = np.zeros((n_tiles * 32, 24))
    W[:6890] = np.asarray(model['weights'], np.float64)
    ref = A.astype(np.float64) @ W.T
    got = T / (1024.0 * 16384.0)
    assert np.abs(got[:, :6890] - ref[:, :6890]).max() < 3e-6 * np.abs(ref).max()


def test_resnet_class_surface_matches_the_reference():
    """models/resnet.py:39-145: BasicBlock / Bottleneck / ResNet(block, layers, in_channels, num_classes=1000, zero_init_residual=False, groups=1,
    width_per_group=64, replace_stride_with_dilation=None, norm_layer=None) -- the class surface behind the factories (missing
    #5).  Same constructor signatures, same state-dict keys and seeded initialisation as the factories; what no kernel implements raises."""
    import inspect
    import straps_amd
    from straps_amd.resnet import ResNet, BasicBlock, Bottleneck
    assert BasicBlock.expansion == 1 and Bottleneck.expansion == 4
    want = ['self', 'inplanes', 'planes', 'stride', 'downsample', 'groups', 'base_width', 'dilation', 'norm_layer']
    assert list(inspect.signature(BasicBlock.__init__).parameters) == want and list(inspect.signature(Bottleneck.__init__).parameters) == want
    assert list(inspect.signature(ResNet.__init__).parameters)[:10] == ['self', 'block', 'layers', 'in_channels', 'num_classes', 'zero_init_residual', 'groups',
                                                                         'width_per_group', 'replace_stride_with_dilation', 'norm_layer']
    for block, layers, factory in ((BasicBlock, [2, 2, 2, 2], straps_amd.resnet18), (Bottleneck, [3, 4, 6, 3], straps_amd.resnet50)):
        torch.manual_seed(5)
        a = ResNet(block, layers, 18)                       # positional, like models/resnet.py:219
        torch.manual_seed(5)
        b = factory(18)
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb)
        assert all(torch.equal(sa[k], sb[k]) for k in sa)
        assert [n for n, _ in a.named_parameters()] == [n for n, _ in b.named_parameters()]
        assert all(isinstance(u, block) for li in range(1, 5) for u in getattr(a, 'layer%d' % li))
    z = ResNet(Bottleneck, [1, 1, 1, 1], 3, num_classes=10, zero_init_residual=True)
    assert float(z.layer1[0].bn3.weight.abs().sum()) == 0.0 and float(z.layer1[0].bn2.weight.abs().sum()) > 0
    blk = BasicBlock(64, 128, 2, torch.nn.Sequential(torch.nn.Conv2d(64, 128, 1, 2, bias=False), torch.nn.BatchNorm2d(128)))
    assert list(blk.state_dict())[-7] == 'downsample.0.weight' or 'downsample.0.weight' in blk.state_dict()
    with pytest.raises(ValueError):
        BasicBlock(64, 64, groups=2)                        # models/resnet.py:49-50
    with pytest.raises(NotImplementedError):
        BasicBlock(64, 64, dilation=2)                      # :51-52
    with pytest.raises(NotImplementedError):
        ResNet(Bottleneck, [3, 4, 6, 3], 18, groups=32, width_per_group=4)
    with pytest.raises(ValueError):
        ResNet(BasicBlock, [2, 2, 2, 2], 18, replace_stride_with_dilation=[False])
    with pytest.raises(NotImplementedError):
        ResNet(BasicBlock, [2, 2, 2, 2], 18, norm_layer=torch.nn.GroupNorm)
    assert ResNet('basic', [2, 2, 2, 2], 18).kind == 'basic'                      # (the spelling of earlier rounds)
