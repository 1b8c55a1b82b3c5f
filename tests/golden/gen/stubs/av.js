// Minimal stand-in for the Aurora.js (`av`) peer dependency so that the reference's
// src/decoder.js and src/adts_demuxer.js can be require()'d without the framework.
// Only the class plumbing is provided (extend/register); no arithmetic lives in `av`.
function makeBase() {
    function Base() {}
    Base.extend = function (body) {
        function Klass() {}
        Klass.prototype = Object.create(Base.prototype);
        Klass.extend = Base.extend;
        Klass.register = Base.register;
        body.call(Klass, Klass);
        return Klass;
    };
    Base.register = function () {};
    return Base;
}
exports.Decoder = makeBase();
exports.Demuxer = makeBase();
exports.Stream = {};
exports.Bitstream = function () {};
exports.Buffer = function (data) { this.data = data; };
