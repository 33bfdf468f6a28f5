"""get_encoder -- mirror of reconstruction/encoding.py:45-96 restricted to the encoders on the hot path
('triplane_wavelet' for positions, 'sphere_harmonics' for directions, 'None')."""


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=2048, align_corners=False, bound=1, **kwargs):
    if encoding == 'None':
        return lambda x, **kwargs: x, input_dim
    elif encoding == 'sphere_harmonics':
        from .shencoder import SHEncoder
        encoder = SHEncoder(input_dim=input_dim, degree=degree)
    elif encoding == 'triplane_wavelet':
        from .triplaneencoder.triplane_encoder import TriPlaneVolume
        extra = {}
        if 'plane_dtype' in kwargs:
            extra['plane_dtype'] = kwargs['plane_dtype']
        encoder = TriPlaneVolume(  # reference: encoding.py:76-93
            number_of_features=kwargs['triplane_channels'],
            plane_resolution=kwargs['triplane_resolution'],
            init_sigma=0.1,
            lbound=bound,
            viewdir_plane_resolution=-1,
            apply_activation_on_features=False,
            inner_multi_res_scale=kwargs['triplane_wavelet_levels'],
            inner_multi_res_scale_current=1,
            learn_rotation_axis=kwargs.get('learn_rotation_axis', False),
            dropout=kwargs.get('dropout', 0),
            wavelet_type=kwargs.get('wavelet_type', 'bior6.8'),
            lbound_auto_scale=kwargs.get('lbound_auto_scale', False),
            upscale_ratio_bound=kwargs.get('upscale_ratio_bound', -1),
            upscale_levels=kwargs.get('upscale_levels', 2),
            wavelet_base_resolution=kwargs.get('wavelet_base_resolution', 0),
            **extra,
        )
    else:
        raise NotImplementedError(
            f"encoding `{encoding}` is not part of the TriNeRFLet hot path (frequency/hashgrid/tiledgrid/ash are "
            "out of scope, SURVEY.md 2.1)")
    return encoder, encoder.output_dim
